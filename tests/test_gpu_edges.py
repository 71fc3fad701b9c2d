"""GPU: edge cases of the three bands through the C ABI - smallest / odd / large frames, batches that are not a multiple
of max_batch, argument errors.  The reference has no tests of its own; these pin the behaviour the band scripts rely on."""
import numpy as np
import pytest

from prisma_amd import engine, synth

pytestmark = pytest.mark.gpu
Err = engine._lib.PrismaBandsError


@pytest.fixture(scope="module")
def depth():
    net = engine.DepthAnything(synth.depth_anything_weights("vits", seed=1234), "vits", max_batch=3)
    yield net
    net.close()


def test_depth_tiny_odd_and_4k_frames(depth):
    for h, w in ((1, 1), (7, 13), (33, 517), (2160, 3840)):
        fr = synth.frames(1, h, w, seed=h + w)
        d, rgb, mn, mx = depth.infer_batch(fr)
        assert d.shape == (1, h, w) and rgb.shape == (1, h, w, 3) and np.isfinite(d).all()
        assert mn[0] == d.min() and mx[0] == d.max()
        if mx[0] > mn[0]:
            hot = np.unravel_index(d[0].argmax(), d[0].shape)
            assert tuple(rgb[0][hot]) == (0, 25, 255)         # flipped relative depth: the nearest (largest) value is heat 0
        else:                                                 # constant map: 0/0 -> NaN -> byte 0 everywhere (encode.py:13-33)
            assert not rgb.any()


def test_depth_batch_not_multiple_of_max_batch(depth):
    fr = synth.frames(7, 60, 100, seed=9)                     # max_batch 3 -> chunks of 3, 3, 1
    d, rgb, mn, mx = depth.infer_batch(fr)
    one, rgb1, mn1, mx1 = depth.infer_batch(fr[6:7])
    assert np.allclose(d[6], one[0], rtol=0, atol=2e-3 * float(one.max())) and abs(mn[6] - mn1[0]) <= 2e-3 * mx1[0]
    assert (np.abs(rgb[6].astype(int) - rgb1[0].astype(int)) > 2).mean() < 1e-2


def test_depth_argument_errors(depth):
    with pytest.raises(Err):
        depth.infer_batch(np.zeros((0, 8, 8, 3), np.uint8))
    with pytest.raises(AssertionError):
        depth.infer_batch(np.zeros((1, 8, 8, 4), np.uint8))
    with pytest.raises(Err):
        engine.DepthAnything({"pretrained.cls_token": np.zeros((1, 1, 384), np.float32)}, "vits")     # missing weights


def test_flow_minimum_sequence_and_too_small_frames():
    net = engine.FlowRaft(synth.raft_weights(seed=4321))
    fr = synth.frame_pair_sequence(2, 128, 128, seed=1)       # smallest legal: the 4-level pyramid needs 16 x 16 at 1/8
    flow, rgb, mx = net.infer_sequence(fr, scale=1.0, iters=1)
    assert flow.shape == (1, 1, 128, 128, 2) and np.isfinite(flow).all()
    with pytest.raises(Err, match="too small"):               # the reference returns NaN here (level-3 map of height 1)
        net.infer_sequence(synth.frame_pair_sequence(2, 96, 160, seed=1), scale=1.0, iters=1)
    with pytest.raises(AssertionError):
        net.infer_sequence(fr[:1], scale=1.0, iters=1)
    with pytest.raises(Err):
        net.infer_sequence_masks(fr, scale=1.0, iters=0)
    # a sequence longer than anything planned before re-plans the arena
    fr9 = synth.frame_pair_sequence(9, 128, 136, seed=2)
    flow, _, mx = net.infer_sequence(fr9, scale=1.0, iters=2, backward=True, want_rgb=False)
    assert flow.shape == (8, 2, 128, 136, 2) and (mx > 0).all()
    net.close()


def test_mask_small_frame_and_no_detection():
    cfg = synth.MASK_CFGS["tiny"]
    net = engine.MaskMMDet(synth.solov2_weights(cfg), cfg, max_batch=2)
    keep = [synth.COCO_CLASSES.index(c) for c in synth.BAND_CLASSES]
    out = net.infer_batch(synth.frames(3, 37, 53, seed=4), 0.5, keep)          # upscaled to the 320 x 192 test scale
    assert out.shape == (3, 37, 53, 3)
    none = net.infer_batch(np.zeros((1, 64, 64, 3), np.uint8), 1.1, keep)      # nothing can pass confidence 1.1
    assert not none.any() and len(net.instances(0)[0]) >= 0
    with pytest.raises(Err):
        net.infer_batch(np.zeros((0, 8, 8, 3), np.uint8))
    net.close()


@pytest.mark.gpu
def test_rccl_gather_scalars_through_the_c_abi():
    """pb_comm_unique_id / pb_comm_init / pb_gather_scalars on a one-rank RCCL communicator: the library opens
    librccl itself, builds the communicator on the ctx's GPU and runs ncclAllGather on the ctx stream (with one rank the
    gathered block is the local block).  The N > 1 ordering logic around it is covered on CPU (tests/test_shard_cpu.py)."""
    from prisma_amd import engine
    ops = engine.Ops()
    cid = ops.comm_unique_id()
    assert len(cid) == 128 and any(cid)
    ops.comm_init(cid, 0, 1)
    x = np.arange(24, dtype=np.float32).reshape(12, 2) * 0.5
    g = ops.gather_scalars(x)
    assert g.shape == (1, 12, 2) and np.array_equal(g[0], x)
    g2 = ops.gather_scalars(np.float32([[3.0, 4.0, 5.0]]))           # a smaller payload afterwards reuses the buffer
    assert np.array_equal(g2[0], np.float32([[3.0, 4.0, 5.0]]))
    with pytest.raises(engine._lib.PrismaBandsError, match="already has a communicator"):
        ops.comm_init(cid, 0, 1)
    ops.close()


@pytest.mark.gpu
def test_bench_nccl_path_under_torch_distributed_run():
    """bench.py exactly as the driver launches it for N > 1 (python -m torch.distributed.run ... bench.py --gpus N), with one rank
    and PRISMA_FORCE_DIST=1 so that the "nccl" (= RCCL) process group, its barrier, the max all-reduce and the scalar
    all_gather_into_tensor really execute on this single-GPU box; tiny sizes."""
    import json, os, socket, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, PRISMA_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "1", "--batch", "3",
           "--height", "270", "--width", "480", "--encoder", "vits", "--no-cpu-baseline", "--one-precision"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["value"] > 0 and line["scaling"] == "weak" and "roofline" in line


def test_depth_host_pipeline_with_page_locked_buffers(depth):
    """pb_depth_infer_batch reads page-locked caller frames and writes page-locked result arrays directly (no staging copy); chunked by
    the host_chunk option.  Same bytes as the pageable path."""
    import torch
    fr = synth.frames(5, 90, 120, seed=3)
    d0, r0, mn0, mx0 = depth.infer_batch(fr)
    depth.set_option("host_chunk", 2)
    try:
        hf = torch.from_numpy(fr).pin_memory()
        od = torch.empty(d0.shape, dtype=torch.float32).pin_memory()
        og = torch.empty(r0.shape, dtype=torch.uint8).pin_memory()
        d1, r1, mn1, mx1 = depth.infer_batch(hf.numpy(), out_depth=od.numpy(), out_rgb=og.numpy())
        assert d1.ctypes.data == od.data_ptr()
        assert np.array_equal(d1, d0) and np.array_equal(r1, r0) and np.array_equal(mn1, mn0) and np.array_equal(mx1, mx0)
        d2, r2, _, _ = depth.infer_batch(fr)                     # pageable again, chunked
        assert np.array_equal(d2, d0) and np.array_equal(r2, r0)
    finally:
        depth.set_option("host_chunk", 0)


def test_bands_run_concurrently_equal_sequential():
    """engine.run_concurrently (the bench's timed region) and the two-host-thread form of the same thing: depth_anything and flow_raft on
    their own ctx streams at once produce the bytes of running them one after the other - contexts share no buffers, and a frame's result
    does not depend on what else the GPU is doing.  Replaces the reference's strictly sequential band order (process.py:205-290)."""
    import threading
    torch = pytest.importorskip("torch")
    B, H, W = 6, 360, 640
    frames = synth.frame_pair_sequence(B, H, W, seed=77)
    dn = engine.DepthAnything(synth.depth_anything_weights("vits", seed=1234), "vits", max_batch=B)
    fn = engine.FlowRaft(synth.raft_weights(seed=4321))
    d_frames = torch.from_numpy(frames).cuda()
    sh, sw = engine.flow_out_size(H, W, 0.75)

    def buffers():
        return (torch.zeros((B, H, W, 3), dtype=torch.uint8, device="cuda"), torch.zeros((B - 1, sh, sw, 3), dtype=torch.uint8, device="cuda"),
                torch.zeros((3, B), dtype=torch.float32, device="cuda"))

    def jobs(d_rgb, f_rgb, scal):
        def depth_job():
            dn.infer_dev(d_frames.data_ptr(), B, H, W, 0, d_rgb.data_ptr(), scal[0].data_ptr(), scal[1].data_ptr(), True)

        def flow_job():
            fn.infer_sequence_dev(d_frames.data_ptr(), B, H, W, 0.75, 6, False, 0, f_rgb.data_ptr(), scal[2].data_ptr())
        return depth_job, flow_job

    ref = buffers()
    dj, fj = jobs(*ref)
    dj(); dn.sync(); fj(); fn.sync()
    ref = [t.cpu().numpy() for t in ref]
    assert ref[0].any() and ref[1].any() and (ref[2][1] > ref[2][0]).all()
    for mode in ("run_concurrently", "threads"):
        for rep in range(3):
            out = buffers()
            dj, fj = jobs(*out)
            if mode == "run_concurrently":
                done = engine.run_concurrently([(dn, dj), (fn, fj)])
                assert len(done) == 2 and all(t > 0 for t in done)
            else:
                def run(job, net):
                    job(); net.sync()
                th = threading.Thread(target=run, args=(fj, fn))
                th.start()
                run(dj, dn)
                th.join()
            for a, b in zip(ref, out):
                assert np.array_equal(a, b.cpu().numpy()), (mode, rep)
    dn.close(); fn.close()


def test_submit_wait_streams_clips_with_the_bytes_of_the_blocking_calls():
    """pb_depth_submit_batch / pb_flow_submit_sequence / pb_wait: asynchronous host-pointer calls on page-locked buffers, two submissions in flight
    per context (the second one's uploads under the first one's kernels) - three different clips streamed through both bands give, clip by clip,
    the bytes of the blocking entry points; pageable buffers are refused; pb_wait without a submission is an error."""
    torch = pytest.importorskip("torch")
    B, H, W = 6, 200, 328
    dn = engine.DepthAnything(synth.depth_anything_weights("vits", seed=1234), "vits", max_batch=B)
    fn = engine.FlowRaft(synth.raft_weights(seed=4321))
    dn.set_option("host_chunk", 2); fn.set_option("host_chunk", 2)          # three chunks per call: slot reuse inside a submission as well
    sh, sw = engine.flow_out_size(H, W, 1.0)
    clips = [synth.frame_pair_sequence(B, H, W, seed=80 + k) for k in range(3)]
    ref = []
    for fr in clips:
        d, rgb, mn, mx = dn.infer_batch(fr)
        fl, frgb, fmx = fn.infer_sequence(fr, scale=1.0, iters=3)
        ref.append((d, rgb, mn, mx, fl, frgb, fmx))
    pin = lambda shape, dt: torch.empty(shape, dtype=dt).pin_memory()      # noqa: E731
    sets = [dict(fr=pin((B, H, W, 3), torch.uint8), d=pin((B, H, W), torch.float32), rgb=pin((B, H, W, 3), torch.uint8), mn=pin((B,), torch.float32),
                 mx=pin((B,), torch.float32), fl=pin((B - 1, 1, sh, sw, 2), torch.float32), frgb=pin((B - 1, 1, sh, sw, 3), torch.uint8),
                 fmx=pin((B - 1, 1), torch.float32)) for _ in range(2)]

    def submit(k):
        s = sets[k % 2]
        s["fr"].copy_(torch.from_numpy(clips[k]))
        for key in ("d", "rgb", "mn", "mx", "fl", "frgb", "fmx"):
            s[key].zero_()
        dn.submit_batch(s["fr"].numpy(), out_rgb=s["rgb"].numpy(), out_depth=s["d"].numpy(), out_min=s["mn"].numpy(), out_max=s["mx"].numpy())
        fn.submit_sequence(s["fr"].numpy(), scale=1.0, iters=3, out_flow=s["fl"].numpy(), out_rgb=s["frgb"].numpy(), out_max=s["fmx"].numpy())

    def check_clip(k):
        d, rgb, mn, mx = dn.wait()
        fl, frgb, fmx = fn.wait()
        for got, want in zip((d, rgb, mn, mx, fl, frgb, fmx), ref[k]):
            assert np.array_equal(got, want), k
    submit(0)
    submit(1)                       # two submissions in flight on each context
    check_clip(0)
    submit(2)                       # re-uses clip 0's host buffers and both device slots while clip 1 is still running
    check_clip(1)
    check_clip(2)
    with pytest.raises(Err):
        dn.wait()
    with pytest.raises(Err, match="page-locked"):
        dn.submit_batch(clips[0], out_rgb=sets[0]["rgb"].numpy())
    d, rgb, mn, mx = dn.infer_batch(clips[1])            # the blocking call after the asynchronous ones
    assert np.array_equal(d, ref[1][0]) and np.array_equal(rgb, ref[1][1])
    submit(0)                                            # ... and a blocking call WHILE a submission is in flight: it queues behind it on the device
    d, rgb, mn, mx = dn.infer_batch(clips[2])
    fl, frgb, fmx = fn.infer_sequence(clips[2], scale=1.0, iters=3)
    assert np.array_equal(d, ref[2][0]) and np.array_equal(rgb, ref[2][1]) and np.array_equal(fl, ref[2][4]) and np.array_equal(fmx, ref[2][6])
    check_clip(0)
    dn.close(); fn.close()
