#!/usr/bin/env python3
"""bench.py -- images/s of the yolov3-tiny INT8 path on MI355X (BASELINE.json metric), with the dominant kernel's
roofline and the reference's CPU path timed beside it.

  python bench.py --gpus N --steps K --warmup W          (N>1: launched by torch.distributed.run, one rank per GPU)

A step = one pass of the hot path (the plain-C darknet host's forward_network_gpu: input layout conversion + 24
layer.forward_gpu launches) over one batch of 64 synthetic uint8 416x416 images that are already resident in HBM in
the reference's [B][C][H][W] layout.  Images shard embarrassingly over the ranks (weak scaling, 64 per GPU); the only
collective is a one-time RCCL broadcast of the packed quantized weights at start-up.  Rank 0 prints ONE JSON line.
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_INT8_TOPS = 256 * 4 * 2048 * 2.4e9 / 1e12  # 256 CUs x 4 SIMDs x 2048 int8 ops/clk/SIMD x 2.4 GHz = 5033 TOP/s dense
PEAK_HBM_GBS = 8000.0


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=64, help="images per GPU per step")
    ap.add_argument("--cfg", default=os.path.join(ROOT, "cfg", "yolov3-tiny_quant.cfg"))
    ap.add_argument("--graph", action="store_true", help="replay the layer loop as a hipGraph (no per-layer events)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-images", type=int, default=8, help="bounded CPU-baseline sample (images through the reference)")
    ap.add_argument("--layers", action="store_true", help="print the per-layer table to stderr")
    ap.add_argument("--cpu-omp", action="store_true", help="also time the reference's OpenMP build on all host cores (extra JSON key)")
    ap.add_argument("--no-ref-f32", action="store_true", help="skip the extra leg that times the bit-faithful MI355_ACC_REF_F32 mode")
    ap.add_argument("--ref-f32-steps", type=int, default=2)
    ap.add_argument("--selfcheck-passes", type=int, default=48,
                    help="determinism self-check before the warmup steps: that many passes over the input, yolo-output checksums compared (0: off)")
    return ap.parse_args()


METRICS = {"yolov3-tiny_quant.cfg": "images/sec yolov3-tiny INT8 416x416", "yolov3_quant.cfg": "images/sec yolov3 (full, 75 conv + 23 quantized shortcut) INT8 608x608"}


def conv_layer_work(info, batch):
    """Algorithmic ops / bytes of one conv launch: 2*M*K*N ops, uint8 in + weights + uint8 out."""
    K = info["c"] * info["size"] * info["size"]
    N = info["out_h"] * info["out_w"] * batch
    ops = 2.0 * info["n"] * K * N
    byt = info["c"] * info["h"] * info["w"] * batch + info["n"] * K + info["n"] * N
    return ops, byt


def pmc_traffic_per_launch():
    """HBM bytes per launch of the dominant kernel from the newest COMMITTED rocprofv3 PMC passes
    (profiles/*_pmc_traffic.json, produced by tools/gpu_session.sh pmc + tools/pmc_traffic.py: separate FETCH_SIZE /
    WRITE_SIZE passes, gfx950 FETCH_SIZE x2 correction) -- NOT measured in this run (PMC collection needs rocprofv3
    around the process).  Returns (bytes or None, source file or None)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json")))  # rNN_vM_...: the name orders them (mtimes do not survive a snapshot)
    if not files:
        return None, None
    rows = [r for r in json.load(open(files[-1])) if "conv_rows_i8_kernel" in r["kernel"] or "conv_rows16_i8_kernel" in r["kernel"]]
    n = sum(r["launches"] for r in rows)
    if not n:
        return None, None
    tot = sum(((r["hbm_read_bytes_per_launch"] or 0) + (r["hbm_write_bytes_per_launch"] or 0)) * r["launches"] for r in rows)
    return tot / n, os.path.relpath(files[-1], ROOT)


def cpu_baseline(cfg, wts, nimg, omp=False):
    """The reference itself (oracle/_ref/libdarknet_ref.so, Makefile-default build, 1 thread; omp=True: its MULTI_CORE=1
    OpenMP flavour on every host core) timed on this box's host cores on a bounded sample; falls back to the CPU
    restatement ('port') if the prebuilt reference is absent."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    from yolo_quantization_amd import synth
    _, shapes = synth.layer_shapes(synth.read_cfg(cfg))
    x = synth.synth_image_u8(shapes[0].c, shapes[0].h, shapes[0].w, seed=7)
    if any(L.type == "shortcut" for L in shapes):
        # `[shortcut] quantized=1` is this build's own op: the reference cannot run the net.  Bounded sample: ONE image through
        # the CPU restatement in exact-integer mode, OpenMP over output channels (thread count stated).
        import oracle
        onet = oracle.OracleNet(cfg, wts)
        onet.prepare(np.float32(1.0 / 255.0), 0)
        t0 = time.time()
        onet.forward(x, accum=oracle.ACC_EXACT)
        dt = time.time() - t0
        return {"value": 1 / dt, "unit": "images/s", "cores": int(oracle.lib().orc_omp_threads()), "kind": "port",
                "sample": f"1 x {os.path.basename(cfg)} {shapes[0].h}x{shapes[0].w} image, whole net, batch 1, exact-integer mode "
                          f"(the reference has no quantized [shortcut]), {os.cpu_count()} host cores present"}
    try:
        import refdrv
        if not refdrv.available(omp):
            raise FileNotFoundError("oracle/_ref not built")
        net = refdrv.RefNet(cfg, wts, omp=omp)
        net.prepare(synth.image_u8_to_float(x))
        t0 = time.time()
        for _ in range(nimg):
            net.forward()
        dt = time.time() - t0
        kind = "reference"
    except Exception as e:  # noqa: BLE001
        print(f"[bench] reference CPU baseline unavailable ({e}); timing the oracle restatement instead", file=sys.stderr)
        import oracle
        onet = oracle.OracleNet(cfg, wts)
        onet.prepare(np.float32(1.0 / 255.0), 0)
        nimg = max(1, nimg // 4)
        t0 = time.time()
        for _ in range(nimg):
            onet.forward(x, accum=oracle.ACC_REF_F32)
        dt = time.time() - t0
        kind = "port"
    return {"value": nimg / dt, "unit": "images/s", "cores": (os.cpu_count() if omp and kind == "reference" else 1), "kind": kind,
            "sample": f"{nimg} x {os.path.basename(cfg)} {shapes[0].h}x{shapes[0].w} image, whole net, batch 1, {os.cpu_count()} host cores present"}


def flush_c_stdio():
    """RCCL prints its version banner through C stdio; when stdout is a pipe that text would otherwise appear at process
    exit, after the JSON line."""
    import ctypes
    sys.stdout.flush()
    try:
        ctypes.CDLL(None).fflush(None)
    except OSError:
        pass


def main():
    args = parse_args()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            print(f"[bench] --gpus {args.gpus} needs torch.distributed.run with {args.gpus} ranks", file=sys.stderr)
            sys.exit(2)
    import numpy as np
    import torch  # first: its bundled HIP runtime must be the one the process uses (same SONAME as /opt/rocm's)
    if not torch.cuda.is_available():
        print("[bench] no GPU visible: the INT8 path has no CPU fallback", file=sys.stderr)
        sys.exit(3)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    force_dist = os.environ.get("BENCH_FORCE_DIST") == "1"  # exercise the RCCL start-up path on one rank
    if world > 1 or force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(29500 + os.getppid() % 2000))  # the launcher sets it; this default only serves BENCH_FORCE_DIST
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from yolo_quantization_amd import binding, synth
    binding.init(local_rank)
    if os.environ.get("BENCH_DEBUG_FLAGS"):  # A/B switches between equivalent kernel variants (mi355_debug_flags)
        binding.shim().mi355_debug_flags(int(os.environ["BENCH_DEBUG_FLAGS"]))
    B = args.batch
    wts = f"/tmp/bench_yolov3_tiny_{os.getpid()}.weights"

    # ---- model: rank 0 reads the weights file, preps and packs; the packed bytes travel by RCCL broadcast
    if rank == 0:
        synth.synth_weights(args.cfg, wts, seed=1234)
        net = binding.Net(args.cfg, wts, batch=B, gpu=local_rank, use_graph=args.graph, keep_head_float=False)
        net.prepare_fixed(1.0 / 255.0, 0)
        packed = net.export_packed()
        size_t = torch.tensor([packed.size], dtype=torch.int64, device=dev)
    else:
        net = binding.Net(args.cfg, None, batch=B, gpu=local_rank, use_graph=args.graph, keep_head_float=False)
        size_t = torch.zeros(1, dtype=torch.int64, device=dev)
    bcast_ms = 0.0
    if world > 1 or force_dist:
        dist.broadcast(size_t, 0)
        nbytes = int(size_t.item())
        blob = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        if rank == 0:
            blob.copy_(torch.from_numpy(packed))
        torch.cuda.synchronize()
        t0 = time.time()
        dist.broadcast(blob, 0)
        torch.cuda.synchronize()
        bcast_ms = (time.time() - t0) * 1e3
        flush_c_stdio()
        if rank != 0 or force_dist:
            if force_dist and rank == 0:  # single-rank self test: re-import what was exported
                net.close()
                net = binding.Net(args.cfg, None, batch=B, gpu=local_rank, use_graph=args.graph, keep_head_float=False)
            net.import_packed_gpu(blob.data_ptr(), nbytes)

    if os.environ.get("BENCH_NO_DIRECT_INPUT") == "1":  # A/B: layer 0 through the 4-byte-cell conversion pass instead of reading the planes
        net.set("input_direct", 0)
    # ---- synthetic input, resident in HBM in the reference layout before the timed region
    in_c, in_h, in_w = net.info[0]["c"], net.info[0]["h"], net.info[0]["w"]  # 3 x 416 x 416 for the headline cfg
    x = synth.synth_image_u8(in_c, in_h, in_w, seed=1000 + rank, batch=B)
    net.push_input(x)
    net.sync()

    def barrier():
        if world > 1 or force_dist:
            dist.barrier()
        torch.cuda.synchronize()
        net.sync()

    # ---- determinism self-check (a race detector for kernels scheduled by hand: counted waits, registers reloaded in place): N passes
    # over the resident input, a device-side checksum of the yolo outputs after each; compared after the timed region, a mismatch
    # fails the run.  Nothing is synchronised here, so the passes also leave the device at its steady clocks when the warmup steps
    # start: after any idle phase of >= 20 ms (the host-side set-up above is one) the first ~40 steps of the net run up to 8 %
    # slower (tools/dbg/step_curve.py) -- without this a `--warmup 5 --steps 20` region measures mostly that.
    if args.selfcheck_passes > 0 and not args.graph:
        net.selfcheck(args.selfcheck_passes)
    for _ in range(args.warmup):
        net.forward()
    barrier()
    # per-layer HIP events (launch stream) are recorded on every PROF_STRIDE-th step of the timed region.  An event costs
    # ~3.9 us of stream time: ~100 us for the 26 of a profiled step.  Measured in one box (400 steps): 0.377 ms / step with every
    # 8th step profiled, 0.365 with a single profiled step -- 3 % of `value` went into its own instrumentation.  Every 32nd step
    # (at least one) keeps that below 1 % on long runs (1.4 % at --steps 20).
    # A short timed region also sees the device's start-up: after any idle phase of more than ~2 ms the first ~25 steps run up
    # to 8 % slower (tools/dbg/step_curve.py: 0.400 0.388 0.376 0.370 0.368 .. ms per step in groups of five; --warmup 5
    # --steps 20 gives 0.38, --warmup 50 0.361, --warmup 200 0.355 in the same box).  Nothing is done about that here.
    prof_stride = max(1, int(os.environ.get("BENCH_PROF_STRIDE", "32")))
    # ... counted back from the LAST step of the timed region (the first ones run on a device that is still ramping up)
    prof_phase = (args.steps - 1) % prof_stride
    prof_steps = 0 if args.graph else min((args.steps - 1 - prof_phase) // prof_stride + 1, 64)
    net.profile_begin(prof_steps, prof_stride, prof_phase)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        net.forward()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms_per_step = dt / args.steps * 1e3
    value = world * B * args.steps / dt
    selfcheck = None
    if args.selfcheck_passes > 0 and not args.graph:
        bad = net.selfcheck_result()
        selfcheck = {"passes": args.selfcheck_passes, "passes_differing_from_the_first": bad}
        if bad:
            raise SystemExit(f"bench.py: determinism self-check FAILED: {bad} of {args.selfcheck_passes} passes over the same input gave other yolo outputs")

    # ---- roofline of the dominant kernel (the MFMA implicit-GEMM conv), from HIP events on the launch stream
    roof = None
    layers = []
    if prof_steps and rank == 0:
        nprof, ms = net.profile_read()
        # An event pair with no launch in between (the fused / elided layers: maxpools after fused convs, elided routes, the
        # yolo layers written by their head conv) measures what recording an event costs on this stream; that cost is
        # taken out of every layer's interval (rocprofv3's kernel durations in profiles/ are the cross-check).
        fused = [net.is_fused(i) for i in range(net.n)]
        empty = [float(ms[i + 1]) / nprof for i, inf in enumerate(net.info)
                 if (i > 0 and fused[i - 1]) or inf["type"] in (binding.T_ROUTE, binding.T_YOLO)]
        ev_cost = min(empty) if empty else 0.0
        # ... but an event recorded behind a running kernel is partly processed in that kernel's shadow: subtracting the
        # empty-interval cost from every interval under-reports the layers (their sum would fall short of a step).  The
        # steps without events give the truth for the sum: T_unprofiled = (dt - nprof * T_profiled) / (steps - nprof),
        # and the per-interval cost that makes the profiled steps' intervals add up to it is what gets subtracted.
        if args.steps > nprof > 0:
            t_prof = float(sum(ms)) / nprof                       # ms of one profiled step, events included
            t_unprof = (dt * 1e3 - nprof * t_prof) / (args.steps - nprof)
            ev_cost = min(ev_cost, max((t_prof - t_unprof) / len(ms), 0.0))
        def on_rows_kernel(i, inf):
            """conv_rows16_i8_kernel / conv_rows_i8_kernel launches: 64-byte channel chunks, served by the implicit-GEMM family (the host records
            which kernel family took each conv of the last step: conv_small / conv1x1 / conv_ws3 take the others)."""
            return inf["type"] == binding.T_CONV and inf["c"] % 64 == 0 and net.conv_kernel(i) == 5
        mf_ops = mf_ms = all_ops = all_ms = 0.0
        for i, inf in enumerate(net.info):
            t_ms = max(float(ms[i + 1]) / nprof - ev_cost, 1e-6)
            row = {"i": i, "type": inf["type"], "ms": round(t_ms, 5)}
            if inf["type"] == binding.T_CONV:
                ops, byt = conv_layer_work(inf, B)
                row.update(tops=round(ops / (t_ms * 1e-3) / 1e12, 2), gbs=round(byt / (t_ms * 1e-3) / 1e9, 1),
                           k=inf["size"], c=inf["c"], n=inf["n"], hw=inf["out_h"])
                all_ops += ops
                if on_rows_kernel(i, inf):
                    mf_ops += ops
                    mf_ms += t_ms
            all_ms += t_ms
            layers.append(row)
        # the north-star target is quoted on the 3x3 stride-1 layers: their aggregate MFMA rate, whichever kernel serves them
        s33_ops = sum(conv_layer_work(inf, B)[0] for i, inf in enumerate(net.info)
                      if inf["type"] == binding.T_CONV and inf["size"] == 3 and inf["stride"] == 1 and inf["c"] > 3)
        s33_ms = sum(layers[i]["ms"] for i, inf in enumerate(net.info)
                     if inf["type"] == binding.T_CONV and inf["size"] == 3 and inf["stride"] == 1 and inf["c"] > 3)
        nlaunch = sum(1 for i, inf in enumerate(net.info) if on_rows_kernel(i, inf))
        nconv = sum(1 for inf in net.info if inf["type"] == binding.T_CONV)
        achieved = mf_ops / (mf_ms * 1e-3) / 1e12 if mf_ms else 0.0
        traffic, traffic_src = pmc_traffic_per_launch()
        roof = {"bound": "mfma", "kernel": f"conv_rows16_i8_kernel / conv_rows_i8_kernel (row-image MFMA implicit GEMM on 64-channel chunks, 3x3 on V_MFMA_I32_16X16X64_I8: {nlaunch} of the step's {nconv} conv launches, "
                          f"{100 * mf_ms / all_ms:.0f}% of its time and {100 * mf_ops / all_ops:.0f}% of its operations)",
                "achieved": round(achieved, 2), "peak": round(PEAK_INT8_TOPS, 1), "unit": "TOP/s",
                "frac": round(achieved / PEAK_INT8_TOPS, 4),
                "traffic": traffic if os.path.basename(args.cfg) == "yolov3-tiny_quant.cfg" else None,
                "traffic_unit": "HBM bytes per launch (rocprofv3 PMC: FETCH_SIZE x2 + WRITE_SIZE, separate passes)",
                "traffic_source": f"committed profile {traffic_src} -- not measured in this run" if traffic_src else None,
                "ops_per_launch_avg": mf_ops / max(nlaunch, 1), "ms_per_launch_avg": round(mf_ms / max(nlaunch, 1), 5),
                "conv3x3_s1_aggregate": {"tops": round(s33_ops / (s33_ms * 1e-3) / 1e12, 1) if s33_ms else None,
                                         "frac": round(s33_ops / (s33_ms * 1e-3) / 1e12 / PEAK_INT8_TOPS, 4) if s33_ms else None,
                                         "ms": round(s33_ms, 5), "layers": "every 3x3 stride-1 conv with c > 3"},
                # what a loop of NOTHING but V_MFMA_I32_32X32X32_I8 sustains on this chip depends on the operand bytes (power
                # management lowers the shader clock): 4 760-4 940 TOP/s on zeros, 3 490 on uniform random bytes
                # (tools/ubench/mfma_data_power.hip, profiles/r02_v3_ubench_mfma_data_power.log).  `peak` / `frac` stay nominal.
                "mfma_only_loop_random_operands": {"tops": 3490.0, "frac_of_it": round(achieved / 3490.0, 4),
                                                   "source": "profiles/r02_v3_ubench_mfma_data_power.log -- not measured in this run"},
                "input_layout_ms": round(max(float(ms[0]) / nprof - ev_cost, 0.0), 5),
                "event_overhead_ms": round(ev_cost, 5)}
        if args.layers:
            for r in layers:
                print("[layer]", json.dumps(r), file=sys.stderr)

    # ---- extra leg: what full-tensor bit-exactness against the Makefile-default reference costs.
    # MI355_ACC_REF_F32 = the verification kernel that emulates the reference's sequential fp32 accumulation (one thread per
    # output); same net, same batch, separate instance, never part of `value`.
    ref_f32 = None
    if rank == 0 and world == 1 and not force_dist and not args.no_ref_f32 and os.path.basename(args.cfg).startswith("yolov3-tiny"):
        rnet = binding.Net(args.cfg, wts, batch=B, gpu=local_rank, accum=binding.ACC_REF_F32)
        rnet.prepare_fixed(1.0 / 255.0, 0)
        rnet.push_input(x)
        rnet.forward(); rnet.sync()
        t0 = time.perf_counter()
        for _ in range(args.ref_f32_steps):
            rnet.forward()
        rnet.sync()
        rdt = time.perf_counter() - t0
        rnet.close()
        ref_f32 = {"images_per_s": round(B * args.ref_f32_steps / rdt, 1), "ms_per_step": round(rdt / args.ref_f32_steps * 1e3, 3),
                   "steps": args.ref_f32_steps, "batch": B,
                   "note": "MI355_ACC_REF_F32: bit-identical to the Makefile-default reference on every tensor (fp32-accumulate emulation, "
                           "tests/test_gpu_parity.py::test_yolov3_tiny_416_ref_f32_equals_reference_hashes); verification mode, not the product path"}

    cpu = cpu_omp = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args.cfg, wts, args.cpu_images)
        if args.cpu_omp:
            cpu_omp = cpu_baseline(args.cfg, wts, args.cpu_images * 4, omp=True)

    if rank == 0:
        out = {"metric": METRICS.get(os.path.basename(args.cfg), f"images/sec {os.path.basename(args.cfg)} INT8"), "value": round(value, 1), "unit": "images/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8 x s8 -> int32 (f64 requant)",
               "data": "synthetic",
               "config": {"workload": "yolov3-tiny full net (cfg/yolov3-tiny_quant.cfg, leaky, per-channel quant), "
                                      f"batch {B}/GPU synthetic uint8 {in_h}x{in_w}, inputs resident in HBM (NCHW uint8)"
                                      if os.path.basename(args.cfg) == "yolov3-tiny_quant.cfg" else
                                      f"{os.path.basename(args.cfg)}, batch {B}/GPU synthetic uint8 {in_h}x{in_w}, inputs resident in HBM (NCHW uint8)",
                          "global_batch": world * B, "parallelism": f"image-sharded x{world}, RCCL weight broadcast once",
                          "launch": "hipGraph replay" if args.graph else f"eager; per-layer HIP events on every {prof_stride}th step of the timed region",
                          "weight_broadcast_ms": round(bcast_ms, 3)},
               "roofline": roof, "cpu_baseline": cpu, "accum_ref_f32_mode": ref_f32, "selfcheck": selfcheck}
        if cpu_omp:
            out["cpu_baseline_allcores"] = cpu_omp
        if layers:
            os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
            json.dump({"ms_per_step": ms_per_step, "layers": layers},
                      open(os.path.join(ROOT, "gpurun_out", f"bench_layers_n{world}.json" if os.path.basename(args.cfg) == "yolov3-tiny_quant.cfg"
                                        else f"bench_layers_{os.path.splitext(os.path.basename(args.cfg))[0]}_n{world}.json"), "w"), indent=1)
    try:
        os.remove(wts)
    except OSError:
        pass
    net.close()
    if world > 1 or force_dist:
        dist.barrier()
        dist.destroy_process_group()
    flush_c_stdio()
    if rank == 0:  # the ONE JSON line, last on stdout (RCCL's start-up banner sits in C stdio's buffer until flushed)
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
