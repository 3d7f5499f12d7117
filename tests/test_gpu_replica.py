"""network_replica (batches in flight): a replica runs the parent's model bit for bit, alone and concurrently with it."""
import os
import numpy as np
import pytest

from yolo_quantization_amd import binding, synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _outputs(net):
    out = []
    for i, inf in enumerate(net.info):
        if inf["type"] == binding.T_YOLO:
            out.append(net.pull(i)["f32"].copy())
    return out


@pytest.mark.parametrize("cfg,batch,hw", [("tiny_unit.cfg", 2, None), ("yolov3-tiny_quant.cfg", 8, None)])
def test_replica_equals_parent_and_runs_concurrently(tmp_path, cfg, batch, hw):
    binding.init(0)
    cfgp = os.path.join(ROOT, "cfg", cfg)
    wts = str(tmp_path / "w.weights")
    synth.synth_weights(cfgp, wts, seed=3)
    parent = binding.Net(cfgp, wts, batch=batch)
    parent.prepare_fixed(1.0 / 255.0, 0)
    c, h, w = parent.info[0]["c"], parent.info[0]["h"], parent.info[0]["w"]
    xs = [synth.synth_image_u8(c, h, w, seed=50 + k, batch=batch) for k in range(4)]
    # reference results: the parent alone, one input after the other
    want = []
    for x in xs:
        parent.push_input(x)
        parent.forward()
        parent.sync()
        want.append(_outputs(parent))
    reps = [parent.replica(), parent.replica(), parent.replica(default_stream=True)]  # the fourth executor on the default stream
    nets = [parent] + reps
    for nk, x in zip(nets, xs):
        nk.push_input(x)
        nk.sync()
    for _ in range(6):  # several rounds queued on four streams without synchronisation in between
        for nk in nets:
            nk.forward()
    for nk in nets:
        nk.sync()
    for k, nk in enumerate(nets):
        got = _outputs(nk)
        assert len(got) == len(want[k]) and len(got) > 0
        for a, b in zip(got, want[k]):
            assert np.array_equal(a, b), f"instance {k} differs from the parent run alone"
    # every quantized tensor too (the replica planned its own fusion / views from the same cfg)
    parent.push_input(xs[1])
    parent.forward()
    parent.sync()
    for i, inf in enumerate(parent.info):
        if inf["type"] == binding.T_YOLO or parent.is_fused(i):
            continue
        assert np.array_equal(parent.pull(i)["u8"], reps[0].pull(i)["u8"]), f"layer {i}"
    for nk in reversed(nets):
        nk.close()


def test_throughput_plan_same_bytes_other_kernels(tmp_path):
    """mi355_conv_desc.plan only chooses kernels: yolov3-tiny at batch 64 under the throughput plan (row-image 128 x 128 tiles
    for the one-round 3x3 layers) stores the bytes of the latency plan (weights-stationary 3x3 kernel, 128 x 384 tiles)."""
    binding.init(0)
    cfgp = os.path.join(ROOT, "cfg", "yolov3-tiny_quant.cfg")
    wts = str(tmp_path / "w.weights")
    synth.synth_weights(cfgp, wts, seed=5)
    x = synth.synth_image_u8(3, 416, 416, seed=77, batch=64)
    res = {}
    for plan in (0, 1):
        net = binding.Net(cfgp, wts, batch=64)
        net.set("plan", plan)
        net.prepare_fixed(1.0 / 255.0, 0)
        net.push_input(x)
        net.forward()
        net.sync()
        kern = {i: net.conv_kernel(i) for i, inf in enumerate(net.info) if inf["type"] == binding.T_CONV}
        tens = {}
        for i, inf in enumerate(net.info):
            if inf["type"] == binding.T_YOLO:
                tens[i] = net.pull(i)["f32"]
            elif not net.is_fused(i):
                tens[i] = net.pull(i)["u8"]
        res[plan] = (kern, tens)
        net.close()
    k0, t0 = res[0]
    k1, t1 = res[1]
    assert k0[8] == 4 and k0[10] == 4 and k0[14] == 4, k0          # conv_ws3 (whole-LDS workgroups) alone on the device
    assert k1[8] == 5 and k1[10] == 5 and k1[14] == 5 and k1[12] == 5 and k1[21] == 5, k1  # row-image kernel, half-CU workgroups
    common = sorted(set(t0) & set(t1))
    assert len(common) >= 12
    for i in common:
        assert np.array_equal(t0[i], t1[i]), f"layer {i}: the plans disagree"


def test_layer_range_reruns_one_layer_in_place(tmp_path):
    """the diagnostic layer range (tools/layer_flood.py, bench.py's sustained leg): forward_network_gpu over [lo, hi) only, on the tensors
    the last full pass left behind, writes the same bytes again and touches nothing else"""
    binding.init(0)
    cfgp = os.path.join(ROOT, "cfg", "yolov3-tiny_quant.cfg")
    wts = str(tmp_path / "w.weights")
    synth.synth_weights(cfgp, wts, seed=9)
    net = binding.Net(cfgp, wts, batch=4)
    net.prepare_fixed(1.0 / 255.0, 0)
    net.push_input(synth.synth_image_u8(3, 416, 416, seed=3, batch=4))
    net.forward()
    net.sync()
    keep = {i: net.pull(i)["u8"].copy() for i, inf in enumerate(net.info) if inf["type"] != binding.T_YOLO and not net.is_fused(i)}
    yolo = {i: net.pull(i)["f32"].copy() for i, inf in enumerate(net.info) if inf["type"] == binding.T_YOLO}
    for lo, hi in ((12, 13), (14, 17), (21, 24), (0, 2)):
        net.set("range_lo", lo); net.set("range_hi", hi)
        for _ in range(2):
            net.forward()
        net.sync()
    net.set("range_lo", 0); net.set("range_hi", 0)
    for i, v in keep.items():
        assert np.array_equal(net.pull(i)["u8"], v), f"layer {i}"
    for i, v in yolo.items():
        assert np.array_equal(net.pull(i)["f32"], v), f"yolo {i}"
    net.close()


def test_stream_pool_hands_out_streams_that_run_side_by_side():
    """mi355_stream_acquire: the first three streams per device come from the measured pool (distinct, reusable after release);
    whatever was created before them -- here two plain streams -- does not change that"""
    import ctypes as C
    import time
    binding.init(0)
    S = binding.shim()
    junk = [C.c_void_p() for _ in range(2)]
    for j in junk:
        binding.check(S.mi355_stream_create(C.byref(j)), "create")
    got = [C.c_void_p() for _ in range(4)]
    for g in got:
        binding.check(S.mi355_stream_acquire(C.byref(g)), "acquire")
    vals = [g.value for g in got]
    assert len(set(vals)) == 4 and all(vals)
    # streams in use elsewhere (another network of this process) may hold pool slots: release ours and take them again
    for g in got:
        binding.check(S.mi355_stream_release(g), "release")
    again = [C.c_void_p() for _ in range(3)]
    for g in again:
        binding.check(S.mi355_stream_acquire(C.byref(g)), "acquire")
    assert len({g.value for g in again}) == 3
    for g in again:
        binding.check(S.mi355_stream_release(g), "release")
    for j in junk:
        binding.check(S.mi355_stream_destroy(j), "destroy")


@pytest.mark.parametrize("c,n,hw,batch,replanned", [(256, 512, 13, 64, True), (512, 1024, 13, 32, True), (128, 256, 76, 16, False),
                                                    (512, 1024, 19, 16, False)])
def test_plan_hint_on_single_convs(c, n, hw, batch, replanned):
    """mi355_conv_desc.plan through the C-ABI on single launches: one-round launches on well-filled row images are re-planned (other
    kernel family or tile, same bytes), launches of several rounds and badly filled row images (19-wide maps) keep their kernel"""
    C = binding.C
    S = binding.shim()
    binding.init(0)
    rng = np.random.default_rng(c + n + hw)
    x = rng.integers(0, 256, (batch, c, hw, hw), dtype=np.uint8)
    K = c * 9
    wq = rng.integers(0, 256, (n, K), dtype=np.uint8)
    zp_w = rng.integers(100, 157, n, dtype=np.uint8)
    bias = rng.integers(-2000, 2000, n).astype(np.int32)
    mv = np.full(n, 0.75); sv = np.full(n, 2.0 ** -13)
    xt = binding.DevTensor.from_nchw(x, 0)
    blob = binding.DevBuf.from_numpy(binding.conv_pack(wq, zp_w, c, 3, bias, mv, sv))
    out, fam = {}, {}
    for plan in (0, 1):
        y = binding.DevTensor(batch, hw, hw, n, 23)
        d = binding.ConvDesc(n, c, 3, 1, 1, binding.ACT["leaky"], binding.STORE_WRAP, binding.ACC_EXACT, 0, 23, 1.0, plan)
        binding.check(S.mi355_conv_forward(C.byref(d), xt.ref(), blob.ptr, None, None, y.ref(), None, None, None), "conv")
        binding.check(S.mi355_stream_sync(None), "sync")
        fam[plan] = S.mi355_last_conv_kernel()
        out[plan] = y.to_nchw()
    assert np.array_equal(out[0], out[1]), "the plans disagree"
    if c in (128, 256):
        assert fam[0] == 4                      # conv_ws3 alone on the device
        assert fam[1] == (5 if replanned else 4)
    else:
        assert fam[0] == 5 and fam[1] == 5      # row-image kernel either way (tile 128 x 384 or 128 x 128)


def test_parent_returns_to_the_latency_plan_with_its_fused_pools(tmp_path):
    """under the throughput plan conv_ws3 declines the one-round 128 / 256-channel layers, the fused conv + pool call fails
    and the host clears fuse_next_pool for that layer.  When the last replica is freed (or the plan is set back by hand) the parent
    must run the round 1-2 kernels again -- fused pools included -- and not conv + maxpool as two launches for the rest of its life."""
    binding.init(0)
    cfgp = os.path.join(ROOT, "cfg", "yolov3-tiny_quant.cfg")
    wts = str(tmp_path / "w.weights")
    synth.synth_weights(cfgp, wts, seed=9)
    x = synth.synth_image_u8(3, 416, 416, seed=31, batch=64)
    net = binding.Net(cfgp, wts, batch=64)
    net.prepare_fixed(1.0 / 255.0, 0)
    net.push_input(x)

    def state():
        net.forward()
        net.sync()
        return ({i: net.conv_kernel(i) for i, inf in enumerate(net.info) if inf["type"] == binding.T_CONV},
                {i: net.fuses_next(i) for i in range(net.n)}, _outputs(net))

    k0, f0, y0 = state()
    assert f0[8] and f0[10] and k0[8] == 4 and k0[10] == 4  # conv_ws3 with its fused pools
    rep = net.replica()
    k1, f1, y1 = state()
    assert k1[8] == 5 and k1[10] == 5 and not f1[8] and not f1[10]  # row-image 128 x 128 tiles, stand-alone pools
    rep.close()
    k2, f2, y2 = state()
    assert k2 == k0 and f2 == f0, "the parent did not return to the latency plan's launches"
    net.set("plan", 1)
    k3, f3, y3 = state()
    net.set("plan", 0)
    k4, f4, y4 = state()
    assert k3 == k1 and k4 == k0 and f4 == f0
    for y in (y1, y2, y3, y4):
        for a, b in zip(y, y0):
            assert np.array_equal(a, b)
    net.close()


def test_layer_range_refuses_what_it_cannot_run(tmp_path):
    """the layer-range diagnostic dies with a message (error(), like the reference) instead of reading unwritten tensors"""
    import subprocess
    import sys
    code = f"""
import sys; sys.path.insert(0, {ROOT!r})
from yolo_quantization_amd import binding, synth
binding.init(0)
cfg = {os.path.join(ROOT, 'cfg', 'yolov3-tiny_quant.cfg')!r}
synth.synth_weights(cfg, {str(tmp_path / 'w.weights')!r}, seed=1)
net = binding.Net(cfg, {str(tmp_path / 'w.weights')!r}, batch=2)
net.prepare_fixed(1.0 / 255.0, 0)
net.push_input(synth.synth_image_u8(3, 416, 416, seed=1, batch=2))
net.forward(); net.sync()
net.set("range_lo", 1); net.set("range_hi", 3)   # layer 0's own tensor is never stored (fused with its pool)
net.forward()
"""
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and "layer range" in r.stderr
