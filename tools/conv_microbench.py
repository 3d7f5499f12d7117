#!/usr/bin/env python3
"""One-kernel microbench of the MFMA implicit-GEMM conv through the C-ABI (BASELINE config[1] by default:
3x3 s1 conv 256->256, 52x52, batch 32).  Prints achieved TOP/s from HIP events on the launch stream.

  python tools/conv_microbench.py [--c 256 --n 256 --hw 52 --batch 32 --k 3] [--iters 20] [--tile BM BN] [--mode flat|patch]
  python tools/conv_microbench.py --net       # every conv shape of yolov3-tiny at batch 64
"""
import argparse
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, ROOT)
from yolo_quantization_amd import binding  # noqa: E402

PEAK = 256 * 4 * 2048 * 2.4e9 / 1e12
SHIFT = 13  # requantised values of random data then stay in the few-hundred range, like a real net's

NET = [(16, 32, 208, 3), (32, 64, 104, 3), (64, 128, 52, 3), (128, 256, 26, 3), (256, 512, 13, 3), (512, 1024, 13, 3),
       (1024, 256, 13, 1), (512, 30, 13, 1), (256, 128, 13, 1), (384, 256, 26, 3), (256, 30, 26, 1)]


def run(c, n, hw, k, batch, iters, tile=None, mode=None, check=False, nt=0):
    S = binding.shim()
    rng = np.random.default_rng(1)
    x = rng.integers(0, 256, (batch, c, hw, hw), dtype=np.uint8)
    wq = np.random.default_rng(2).integers(0, 256, (n, c * k * k), dtype=np.uint8)
    if os.environ.get("UBENCH_DATA") == "const":   # constant operands: the matrix pipe toggles little -> is the kernel power / clock bound?
        x[:] = 128; wq[:] = 128
    elif os.environ.get("UBENCH_DATA") == "small":  # low-entropy operands
        x = (x & 3).astype(np.uint8) + 126; wq = (wq & 3).astype(np.uint8) + 126
    zp_w = np.random.default_rng(3).integers(100, 157, n, dtype=np.uint8)
    bias = np.zeros(n, np.int32)
    mv = np.full(n, 0.75); sv = np.full(n, 2.0 ** -SHIFT)
    xt = binding.DevTensor.from_nchw(x, 0)
    y = binding.DevTensor(batch, hw, hw, n, 23)
    blob = binding.DevBuf.from_numpy(binding.conv_pack(wq, zp_w, c, k, bias, mv, sv))
    d = binding.ConvDesc(n, c, k, 1, k // 2, binding.ACT["leaky"], 0, 0, 0, 23, 1.0)
    if tile or mode:
        bm, bn = tile if tile else (0, 0)
        bm |= nt << 16
        flag = (1 << 30) if mode == "patch" else ((1 << 29) if mode == "flat" else 0)
        S.mi355_conv_set_tile(bm, bn | flag)
    st = C.c_void_p(); binding.check(S.mi355_stream_create(C.byref(st)))
    e0 = C.c_void_p(); e1 = C.c_void_p()
    S.mi355_event_create(C.byref(e0)); S.mi355_event_create(C.byref(e1))

    def launch():
        binding.check(S.mi355_conv_forward(C.byref(d), xt.ref(), blob.ptr, None, None, y.ref(), None, None, st), "conv")
    for _ in range(3):
        launch()
    S.mi355_stream_sync(st)
    S.mi355_event_record(e0, st)
    for _ in range(iters):
        launch()
    S.mi355_event_record(e1, st)
    ms = C.c_float()
    binding.check(S.mi355_event_elapsed_ms(e0, e1, C.byref(ms)), "elapsed")
    S.mi355_conv_set_tile(0, 0)
    t = ms.value / iters * 1e-3
    ops = 2.0 * n * c * k * k * hw * hw * batch
    byt = x.size + wq.size + n * hw * hw * batch
    return {"c": c, "n": n, "hw": hw, "k": k, "batch": batch, "us": round(t * 1e6, 2), "tops": round(ops / t / 1e12, 1),
            "frac_mfma_peak": round(ops / t / 1e12 / PEAK, 4), "gbs": round(byt / t / 1e9, 1), "tile": tile, "mode": mode, "nt": nt}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--c", type=int, default=256); ap.add_argument("--n", type=int, default=256)
    ap.add_argument("--hw", type=int, default=52); ap.add_argument("--k", type=int, default=3)
    ap.add_argument("--batch", type=int, default=32); ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--tile", type=int, nargs=2); ap.add_argument("--mode", choices=["flat", "patch"])
    ap.add_argument("--net", action="store_true"); ap.add_argument("--sweep", action="store_true")
    ap.add_argument("--shift", type=int, default=13); ap.add_argument("--plan", action="store_true")
    ap.add_argument("--nt", type=int, default=0); ap.add_argument("--dbg", type=int, default=0, help="mi355_debug_flags value (512 = no chunk rotation)"); ap.add_argument("--timeline", action="store_true"); ap.add_argument("--timeline3", action="store_true"); ap.add_argument("--waveprof", action="store_true")
    ap.add_argument("--ablate", action="store_true", help="timing ablation of the K loop (results are wrong)")
    ap.add_argument("--timeline1", action="store_true", help="conv1x1_ws: per-workgroup phase timestamps (-DMI355_ABLATE build)")
    ap.add_argument("--timeline16", action="store_true", help="conv_rows16: per-workgroup phase timestamps (-DMI355_ABLATE build)")
    ap.add_argument("--timelinem", action="store_true", help="conv_mid_pool: per-workgroup phase timestamps (-DMI355_ABLATE build)")
    a = ap.parse_args()
    SHIFT = a.shift
    binding.init(0)
    if a.dbg:
        binding.shim().mi355_debug_flags(a.dbg)
    if a.ablate:
        S = binding.shim()
        for flags, name in [(0, "full"), (1, "no-dma"), (2, "no-barrier"), (4, "no-mfma"), (16, "no-cellsum"), (3, "no-dma,no-barrier"), (5, "no-dma,no-mfma"), (7, "only-lds-reads"), (15, "nothing-in-loop"), (32, "no-epilogue"), (47, "empty-kernel"), (63, "empty-kernel-no-cellsum"), (11, "only-mfma"), (9, "mfma+barrier"), (3, "lds+mfma"), (15 + 64, "epi-no-requant"), (15 + 128, "epi-no-copyout"), (15 + 64 + 128, "epi-neither")]:
            S.mi355_debug_flags(flags)
            r = run(a.c, a.n, a.hw, a.k, a.batch, a.iters, tuple(a.tile) if a.tile else None, a.mode)
            print(name, r["us"], "us", r["tops"], "TOPS")
        S.mi355_debug_flags(0)
    elif a.timeline1:
        S = binding.shim()
        S.mi355_debug_flags(a.dbg)
        r = run(a.c, a.n, a.hw, a.k, a.batch, 1, None, None)
        S.mi355_stream_sync(None)
        ts = np.zeros((4, 8192), np.int64)
        S.mi355_debug_read_ts1.argtypes = [C.c_void_p]
        assert S.mi355_debug_read_ts1(ts.ctypes.data) == 0
        nb = int((ts[0] > 0).sum())
        t = ts[:, :nb].astype(np.float64) / 100.0
        t0 = t[0].min()
        st = np.sort(t[0] - t0)
        print(f"blocks {nb}; span {t[3].max() - t0:.2f} us; starts: p25 {st[nb // 4]:.2f} p50 {st[nb // 2]:.2f} p75 {st[3 * nb // 4]:.2f} max {st[-1]:.2f}")
        for i, nm in [(0, "pixel tables + image DMA issued"), (1, "parameters, A fragments, everything landed, barrier"), (2, "groups: MFMAs + requantise + stores")]:
            d = t[i + 1] - t[i]
            print(f"  {nm:52s} p50 {np.median(d):6.2f}  min {d.min():6.2f}  max {d.max():6.2f} us")
        d = t[3] - t[0]
        print(f"  {'whole workgroup':52s} p50 {np.median(d):6.2f}  min {d.min():6.2f}  max {d.max():6.2f} us")
    elif a.timelinem:  # conv_mid_pool_kernel (64 -> 128 + pool): needs --pool and the -DMI355_ABLATE build
        S = binding.shim()
        r = run(a.c, a.n, a.hw, a.k, a.batch, 1, None, a.mode)
        S.mi355_stream_sync(None)
        ts = np.zeros((6, 4096), np.int64)
        S.mi355_debug_read_tsm.argtypes = [C.c_void_p]
        assert S.mi355_debug_read_tsm(ts.ctypes.data) == 0
        nb = int((ts[0] > 0).sum())
        t = ts[:, :nb].astype(np.float64) / 100.0
        t0 = t[0].min()
        print(json.dumps(r))
        print(f"blocks {nb}; span first start .. last end {t[5].max() - t0:.2f} us; starts: p50 {np.median(t[0]) - t0:.2f}, p90 {np.percentile(t[0], 90) - t0:.2f}, max {t[0].max() - t0:.2f}")
        for i, nm in [(0, "geometry, image DMA issued"), (1, "parameters to LDS, A fragments issued, pixel tables"), (2, "image / fragments landed, barrier"),
                      (3, "cell sums, box sums (two barriers)"), (4, "groups: MFMA chains + epilogues + stores issued")]:
            d = t[i + 1] - t[i]
            print(f"  {nm:52s} p50 {np.median(d):6.2f}  min {d.min():6.2f}  max {d.max():6.2f} us")
        d = t[5] - t[0]
        print(f"  {'whole workgroup':52s} p50 {np.median(d):6.2f}  min {d.min():6.2f}  max {d.max():6.2f} us")
    elif a.timeline16:
        S = binding.shim()
        S.mi355_debug_flags(a.dbg)
        r = run(a.c, a.n, a.hw, a.k, a.batch, 1, tuple(a.tile) if a.tile else None, a.mode, nt=a.nt)
        S.mi355_stream_sync(None)
        ts = np.zeros((6, 4096), np.int64)
        S.mi355_debug_read_ts16.argtypes = [C.c_void_p]
        assert S.mi355_debug_read_ts16(ts.ctypes.data) == 0
        nb = int((ts[0] > 0).sum())
        t = ts[:, :nb].astype(np.float64) / 100.0  # us
        t0 = t[0].min()
        print(f"blocks {nb}; span first start .. last end {t[5].max() - t0:.2f} us; starts: p50 {np.median(t[0]) - t0:.2f}, p90 {np.percentile(t[0], 90) - t0:.2f}, max {t[0].max() - t0:.2f}")
        for i, nm in [(0, "index math, DMA tables, prologue loads issued"), (1, "accumulator seeds, parameters, first image landed"), (2, "K loop"), (3, "epilogue: box sums + requantise"), (4, "copy-out")]:
            d = t[i + 1] - t[i]
            print(f"  {nm:52s} p50 {np.median(d):6.2f}  min {d.min():6.2f}  max {d.max():6.2f} us")
        d = t[5] - t[0]
        print(f"  {'whole workgroup':52s} p50 {np.median(d):6.2f}  min {d.min():6.2f}  max {d.max():6.2f} us")
    elif a.timeline:  # per-workgroup phase timestamps (needs the -DMI355_ABLATE build)
        S = binding.shim()
        r = run(a.c, a.n, a.hw, a.k, a.batch, 1, tuple(a.tile) if a.tile else None, a.mode, nt=a.nt)
        S.mi355_stream_sync(None)
        ts = np.zeros((6, 4096), np.int64)
        S.mi355_debug_read_ts.argtypes = [C.c_void_p]
        assert S.mi355_debug_read_ts(ts.ctypes.data) == 0
        nb = int((ts[0] > 0).sum())
        t = ts[:, :nb].astype(np.float64) / 100.0  # us
        t0 = t[0].min()
        print(json.dumps(r))
        print(f"blocks {nb}; kernel span first-start..last-end {t[5].max() - t0:.2f} us; start skew p50 {np.median(t[0]) - t0:.2f} max {t[0].max() - t0:.2f}")
        names = ["setup(index math, acc zero, params)", "first DMA wait", "K loop", "epilogue requant", "copy-out"]
        for i, nm in enumerate(names):
            d = t[i + 1] - t[i]
            print(f"  {nm:40s} p50 {np.median(d):6.2f}  min {d.min():6.2f}  max {d.max():6.2f} us")
        d = t[5] - t[0]
        print(f"  {'whole workgroup':40s} p50 {np.median(d):6.2f}  min {d.min():6.2f}  max {d.max():6.2f} us")
        order = np.argsort(t[0])
        late = order[256:] if nb > 256 else []
        if len(late):
            print(f"  second-round blocks: {len(late)}, start p50 {np.median(t[0][late]) - t0:.2f} us")
    elif a.timeline3:  # conv_ws3.hip per-wave phase timestamps (needs the -DMI355_ABLATE build)
        S = binding.shim()
        r = run(a.c, a.n, a.hw, a.k, a.batch, 1, None, None)
        S.mi355_stream_sync(None)
        ts = np.zeros((8, 4096, 8), np.int64)
        S.mi355_debug_read_ts3.argtypes = [C.c_void_p]
        assert S.mi355_debug_read_ts3(ts.ctypes.data) == 0
        nb = int((ts[0, :, 0] > 0).sum())
        t = ts[:, :nb, :].astype(np.float64) / 100.0
        t0 = t[0].min()
        print(json.dumps(r), "last kernel", S.mi355_last_conv_kernel())
        print(f"blocks {nb}; span {t[7].max() - t0:.2f} us; start skew p50 {np.median(t[0]) - t0:.2f} max {t[0].max() - t0:.2f}")
        names = ["stage image (loads + A issue + LDS writes)", "params + tables -> barrier", "box sums -> barrier", "A fragments landed", "phase 1 (partner's groups)", "barrier", "phase 2 / all groups (+ epilogue)"]
        for i, nm in enumerate(names):
            d = t[i + 1] - t[i]
            if (t[i + 1] > 0).all() and (t[i] > 0).all():
                print(f"  {nm:46s} p50 {np.median(d):6.2f}  min {d.min():6.2f}  max {d.max():6.2f} us")
        if hasattr(S, "mi355_debug_read_acc3"):
            ac = np.zeros((8, 4096), np.int64)
            S.mi355_debug_read_acc3.argtypes = [C.c_void_p]
            if S.mi355_debug_read_acc3(ac.ctypes.data) == 0 and ac[6, :nb].max() > 0:
                av = ac[:, :nb].astype(np.float64) / 100.0
                print(f"  persistent form, sums over a workgroup's {av[6].mean() * 100:.1f} tiles (us, thread 0): top-of-tile wait + barrier / staging pass {av[0].mean():.2f} | "
                      f"next tile's DMA issue {av[1].mean():.2f} | cell sums {av[2].mean():.2f} | parameters + tables + barrier {av[3].mean():.2f} | box sums + barrier {av[4].mean():.2f} | "
                      f"group loops {av[5].mean():.2f} | before the first tile {av[7].mean():.2f}")
        print("  phase 4->7:", f"p50 {np.median(t[7] - t[4]):6.2f}  max {(t[7] - t[4]).max():6.2f}")
        d47 = (t[7] - t[4]).mean(axis=1)
        print("  4->7 mean per workgroup, by blockIdx % 8 (XCD):", [round(float(d47[k::8].mean()), 2) for k in range(8)])
        print("  4->7 of workgroups 0..31:", [round(float(v), 1) for v in d47[:32]])
        print("  4->7 per wave index (mean):", [round(float(v), 2) for v in (t[7] - t[4]).mean(axis=0)])
        wp = np.zeros((4096, 8, 4), np.int64)
        S.mi355_debug_read_wp3.argtypes = [C.c_void_p]
        assert S.mi355_debug_read_wp3(wp.ctypes.data) == 0
        w = wp[:nb].astype(np.float64)
        ng = np.full_like(w[:, :, 3], float(os.environ.get("WS3_GROUPS", "6")))
        print("  shader clock over phases 4->7 (cycle counter / wall clock), GHz, by wave:", [round(float(v), 3) for v in (w[:, :, 3] / ((t[7] - t[4]) * 1e3)).mean(axis=0)])
        for k, nm in enumerate(["tap offsets + first 4 B reads", "36-MFMA loop", "epilogue / partial store"]):
            print(f"  shader clocks per group, {nm:32s} mean {np.mean(w[:, :, k] / ng):8.0f}   by wave {[int(v) for v in (w[:, :, k] / ng).mean(axis=0)]}")
        print("  start (t4 - t0min) of workgroups 0..31:", [round(float(v), 1) for v in (t[4].mean(axis=1) - t0)[:32]])
        d = t[7] - t[0]
        print(f"  {'whole wave':46s} p50 {np.median(d):6.2f}  min {d.min():6.2f}  max {d.max():6.2f} us")
    elif a.waveprof:  # per-wave stall profile of the 3x3 K loop (needs the -DMI355_ABLATE build)
        S = binding.shim()
        S.mi355_debug_flags(256 | a.dbg)
        r = run(a.c, a.n, a.hw, a.k, a.batch, 1, tuple(a.tile) if a.tile else None, a.mode, nt=a.nt)
        S.mi355_stream_sync(None)
        S.mi355_debug_flags(0)
        wp = np.zeros((4096, 8, 5), np.int64)
        S.mi355_debug_read_wp.argtypes = [C.c_void_p]
        assert S.mi355_debug_read_wp(wp.ctypes.data) == 0
        nb = int((wp.sum(axis=(1, 2)) > 0).sum())
        w = wp[:nb].astype(np.float64)
        print(json.dumps(r))
        names = ["drain LDS reads of previous step", "s_waitcnt vmcnt (DMA landed)", "s_barrier", "k-half 0 (6 MFMA + DMA issue + 5 reads)", "k-half 1 (6 MFMA + 5 reads)"]
        tot = w.sum(axis=2)
        print(f"blocks {nb}; shader-clock ticks per wave in the K loop: p50 {np.median(tot):.0f}")
        for k, nm in enumerate(names):
            print(f"  {nm:44s} {100 * w[:, :, k].sum() / tot.sum():5.1f} %   per wave p50 {np.median(w[:, :, k]):9.0f}")
        print("  per wave index (mean ticks):", [int(v) for v in w.mean(axis=0).sum(axis=1)])
        for k in range(5):
            print(f"    phase {k} by wave:", [int(v) for v in w[:, :, k].mean(axis=0)])
    elif a.plan:  # rows-kernel tile plan sweep: capacity x tile count, per conv shape of the net
        for c, n, hw, k in NET:
            if c % 64:
                continue
            bm = 128 if n >= 128 else (64 if n > 32 else 32)
            mt = (n + bm - 1) // bm
            total = 64 * hw * hw
            print(json.dumps({"auto": run(c, n, hw, k, 64, a.iters)}))
            for bn in (384, 256, 128):
                if bn == 384 and bm != 128:
                    continue
                nt0 = -(-total // bn)
                cands = {nt0}
                for blocks in (256, 512, 768, 1024, 1536, 2048):
                    if blocks // mt >= nt0:
                        cands.add(blocks // mt)
                for nt in sorted(cands)[:4]:
                    try:
                        r = run(c, n, hw, k, 64, a.iters, (bm, bn), None, nt=nt)
                        print(json.dumps({"c": c, "n": n, "hw": hw, "bn": bn, "nt": nt, "blocks": nt * mt, "us": r["us"], "tops": r["tops"]}))
                    except Exception as e:  # noqa: BLE001
                        print(json.dumps({"c": c, "n": n, "hw": hw, "bn": bn, "nt": nt, "error": str(e)[:80]}))
    elif a.net:
        for c, n, hw, k in NET:
            print(json.dumps(run(c, n, hw, k, 64, a.iters)))
    elif a.sweep:
        for bm, bn in [(128, 256), (128, 128)]:
            for mode in ["flat", "patch"]:
                try:
                    print(json.dumps(run(a.c, a.n, a.hw, a.k, a.batch, a.iters, (bm, bn), mode)))
                except Exception as e:  # noqa: BLE001
                    print(json.dumps({"tile": [bm, bn], "mode": mode, "error": str(e)[:100]}))
    else:
        print(json.dumps(run(a.c, a.n, a.hw, a.k, a.batch, a.iters, tuple(a.tile) if a.tile else None, a.mode, nt=a.nt)))
