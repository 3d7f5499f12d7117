#!/usr/bin/env python3
"""bench.py -- images/s of the yolov3-tiny INT8 path on MI355X (BASELINE.json metric), with the dominant kernel's
roofline and the reference's CPU path timed beside it.

  python bench.py --gpus N --steps K --warmup W          (N>1: launched by torch.distributed.run, one rank per GPU)

A step = one pass of the hot path (the plain-C darknet host's forward_network_gpu: input layout conversion + 24
layer.forward_gpu launches) over one batch of 64 synthetic uint8 416x416 images that are already resident in HBM in
the reference's [B][C][H][W] layout.  Images shard embarrassingly over the ranks (weak scaling, 64 per GPU); the only
collective is a one-time RCCL broadcast of the packed quantized weights at start-up.  Rank 0 prints ONE JSON line.

Batches in flight (--inflight, default 4): the K timed steps are dealt round-robin to that many instances of the prepared
network on the same GPU (network_replica: own activation tensors, own input batch, own HIP stream -- the fourth on the device's
default stream, which owns the fourth hardware queue; ONE copy of the packed weights).  Every step is still one complete forward pass over its own 64 images and all K finish inside the timed region; what
changes is that the device may run kernels of neighbouring steps side by side (one batch's launch gaps, pipeline fills and
VALU-bound first layers under another batch's MFMA-bound layers).  Per-kernel figures (roofline, --layers) are taken in a serial
leg right after the timed region (one batch at a time, the kernel alone on the device) and the serial throughput is reported
next to `value` ("serial"); --inflight 1 makes the timed region itself serial, as in rounds 1 and 2.
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
    ap.add_argument("--cpu-omp", action="store_true", help="(default since round 4; kept for old command lines)")
    ap.add_argument("--no-cpu-omp", action="store_true", help="skip the all-cores leg of the CPU baseline (the reference's MULTI_CORE=1 OpenMP build)")
    ap.add_argument("--cpu-omp-images", type=int, default=4, help="bounded sample of the all-cores leg")
    ap.add_argument("--no-ref-f32", action="store_true", help="skip the extra leg that times the bit-faithful MI355_ACC_REF_F32 mode")
    ap.add_argument("--ref-f32-steps", type=int, default=2)
    ap.add_argument("--inflight-events", action="store_true",
                    help="with batches in flight: also record per-layer events on instance 0 inside the timed region (roofline.in_flight); they "
                         "cost ~0.1 ms of a profiled step and measure launches that share their CUs, so they are off by default")
    ap.add_argument("--no-extra-legs", action="store_true", help="skip the latency-plan leg and the sustained-rate leg (profiling runs)")
    ap.add_argument("--serial-steps", type=int, default=64, help="steps of the serial leg that follows the timed region when --inflight > 1")
    ap.add_argument("--inflight", type=int, default=int(os.environ.get("BENCH_INFLIGHT", "4")),
                    help="batches in flight per GPU: that many network instances (own activations, own HIP stream, same packed weights); "
                         "step i runs on instance i %% inflight, so kernels of consecutive steps overlap on the device")
    ap.add_argument("--dry-dist", action="store_true",
                    help="CPU dry run of the multi-rank start-up (no GPU needed): the same launcher, rendezvous, packed-weights broadcast, image "
                         "sharding and max-over-ranks timing code on the gloo backend, host-only prep; rank 0 prints one JSON line and the "
                         "exit code is 0 only when every rank ends up with byte-identical packed state")
    ap.add_argument("--small-m-channels", type=int, default=0,
                    help="unfriendly model (VERDICT r05 #6): the first N filters of every conv get float weights 64x smaller, i.e. requantisation multipliers ~2e-5 "
                         "that fall outside the kernels' integer requantisation (synth.synth_weights small_m_channels)")
    ap.add_argument("--input", choices=["synthetic", "realimg"], default="synthetic",
                    help="realimg: the real-image fixture tests/golden/realimg_416.npz (network input bytes of a photograph) in every slot of the batch instead of uniform random bytes")
    ap.add_argument("--no-preroll-leg", action="store_true", help="skip the rounds 1-4 style region (W warm-up + K timed steps, no pre-roll) that is timed once ahead of the pre-roll")
    ap.add_argument("--preroll", type=int, default=160,
                    help="untimed in-flight steps queued in front of the W warmup steps (0: none).  The device's power management needs ~100 steps (25 ms) of THIS "
                         "load to settle its clocks after any idle or lightly loaded phase -- the host-side set-up and the self-check passes (whose checksum kernels "
                         "leave the chip mostly idle) are such phases: a 20-step region right behind them measures 0.259-0.265 ms per step where the same region after "
                         "100+ plain steps measures 0.244 (profiles/r05_warmup_curve.log); reported as `preroll_steps`")
    ap.add_argument("--selfcheck-passes", type=int, default=48,
                    help="determinism self-check before the warmup steps: that many passes over the input, yolo-output checksums compared (0: off)")
    return ap.parse_args()


METRICS = {"yolov3-tiny_quant.cfg": "images/sec yolov3-tiny INT8 416x416", "yolov3_quant.cfg": "images/sec yolov3 (full, 75 conv + 23 quantized shortcut) INT8 608x608"}


def conv_layer_work(info, batch):
    """Algorithmic ops of one conv launch: 2*M*K*N, and the UNFUSED byte count uint8 in + weights + uint8 out
    (what the layer would move as a launch of its own; launch_bytes() below is what the launch that really runs has to move)."""
    K = info["c"] * info["size"] * info["size"]
    N = info["out_h"] * info["out_w"] * batch
    ops = 2.0 * info["n"] * K * N
    byt = info["c"] * info["h"] * info["w"] * batch + info["n"] * K + info["n"] * N
    return ops, byt


def launch_bytes(net, i, batch):
    """Algorithmic HBM bytes (read, written) of the launch that serves layer i as the host planned it: input tensor + weights read; only
    the tensors the launch STORES written -- a conv fused with its maxpool / upsample / yolo layer stores that layer's tensor and not
    its own (unless a route also reads it), a fused residual add also reads the `from` tensor.  VERDICT r03: the unfused figure
    counted L0's 177 MB pre-pool tensor, which never leaves the CU."""
    from yolo_quantization_amd import binding
    inf = net.info[i]
    if inf["type"] == binding.T_CONV:
        rd = inf["c"] * inf["h"] * inf["w"] * batch + inf["n"] * inf["c"] * inf["size"] * inf["size"]
        own = inf["n"] * inf["out_h"] * inf["out_w"] * batch
        if net.fuses_next(i):
            nx = net.info[i + 1]
            wr = 0 if net.is_fused(i) else own
            if nx["type"] == binding.T_YOLO:
                wr += 4 * nx["outputs"] * batch            # float activations of the yolo layer
            elif nx["type"] == binding.T_SHORTCUT:
                rd += nx["outputs"] * batch
                wr += nx["outputs"] * batch
            else:                                          # maxpool / upsample
                wr += nx["outputs"] * batch
        else:
            wr = own + (4 * own if inf["quant_stop"] else 0)
        return rd, wr
    if inf["type"] in (binding.T_MAXPOOL, binding.T_UPSAMPLE, binding.T_ROUTE):
        return inf["c"] * inf["h"] * inf["w"] * batch, inf["outputs"] * batch
    if inf["type"] == binding.T_SHORTCUT:
        return 2 * inf["outputs"] * batch, inf["outputs"] * batch
    return 0, 0


def pmc_kernel_rows():
    """rows of the newest COMMITTED rocprofv3 PMC traffic summary (profiles/*_pmc_traffic.json) + its path"""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json")))  # rNN_vM_...: the name orders them
    if not files:
        return [], None
    return json.load(open(files[-1])), os.path.relpath(files[-1], ROOT)


def pmc_traffic_per_launch():
    """HBM bytes per launch of the dominant kernel from the newest COMMITTED rocprofv3 PMC passes
    (profiles/*_pmc_traffic.json, produced by tools/gpu_session.sh pmc + tools/pmc_traffic.py: separate FETCH_SIZE /
    WRITE_SIZE passes, gfx950 FETCH_SIZE x2 correction) -- NOT measured in this run (PMC collection needs rocprofv3
    around the process).  Returns (bytes or None, source file or None)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json")))  # rNN_vM_...: the name orders them (mtimes do not survive a snapshot)
    if not files:
        return None, None
    def avg(names):
        rows = [r for r in json.load(open(files[-1])) if any(nm in r["kernel"] for nm in names)]
        n = sum(r["launches"] for r in rows)
        if not n:
            return None
        return sum(((r["hbm_read_bytes_per_launch"] or 0) + (r["hbm_write_bytes_per_launch"] or 0)) * r["launches"] for r in rows) / n
    rows_only = avg(("conv_rows_i8_kernel", "conv_rows16_i8_kernel"))
    # the north-star layer set's kernels (every 3x3 stride-1 conv with c > 3 under the throughput plan)
    ns = avg(("conv_rows_i8_kernel", "conv_rows16_i8_kernel", "conv_pool16_kernel", "conv_small32_kernel", "conv_small_pool_kernel<32", "conv_mid_pool_kernel"))
    return (ns, rows_only), os.path.relpath(files[-1], ROOT)


def cpu_baseline(cfg, wts, nimg, omp=False, threads=None):
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
        if omp and threads:  # the OpenMP runtime the reference's all-cores build is linked against (GNU libgomp)
            ctypes.CDLL("libgomp.so.1").omp_set_num_threads(int(threads))
        net.prepare(synth.image_u8_to_float(x))
        if omp:
            net.forward()  # thread team start-up is not the reference's per-image cost
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
    return {"value": nimg / dt, "unit": "images/s", "cores": ((threads or os.cpu_count()) if omp and kind == "reference" else 1), "kind": kind,
            "sample": f"{nimg} x {os.path.basename(cfg)} {shapes[0].h}x{shapes[0].w} image, whole net, batch 1, {os.cpu_count()} host cores present"}


def microbench_config1(binding, iters=30):
    """BASELINE config[1]: ONE 3x3 s1 conv 256 -> 256 @52x52, batch 32 (uint8 in / int8 w), launched back to back on one stream through the
    C-ABI; HIP events on that stream around `iters` launches.  No epilogue / HBM excuse here: 102 GOP against ~5 MB of operands."""
    import ctypes as C
    import numpy as np
    S = binding.shim()
    c = n = 256; hw = 52; batch = 32; k = 3
    x = np.random.default_rng(1).integers(0, 256, (batch, c, hw, hw), dtype=np.uint8)
    wq = np.random.default_rng(2).integers(0, 256, (n, c * k * k), dtype=np.uint8)
    zp_w = np.random.default_rng(3).integers(100, 157, n, dtype=np.uint8)
    xt = binding.DevTensor.from_nchw(x, 0)
    y = binding.DevTensor(batch, hw, hw, n, 23)
    blob = binding.DevBuf.from_numpy(binding.conv_pack(wq, zp_w, c, k, np.zeros(n, np.int32), np.full(n, 0.75), np.full(n, 2.0 ** -13)))
    d = binding.ConvDesc(n, c, k, 1, 1, binding.ACT["leaky"], 0, 0, 0, 23, 1.0)
    st = C.c_void_p(); binding.check(S.mi355_stream_create(C.byref(st)), "stream")
    e0 = C.c_void_p(); e1 = C.c_void_p()
    S.mi355_event_create(C.byref(e0)); S.mi355_event_create(C.byref(e1))

    def launch():
        binding.check(S.mi355_conv_forward(C.byref(d), xt.ref(), blob.ptr, None, None, y.ref(), None, None, st), "conv (config[1])")
    for _ in range(5):
        launch()
    S.mi355_stream_sync(st)
    S.mi355_event_record(e0, st)
    for _ in range(iters):
        launch()
    S.mi355_event_record(e1, st)
    ms = C.c_float()
    binding.check(S.mi355_event_elapsed_ms(e0, e1, C.byref(ms)), "elapsed")
    fam = S.mi355_last_conv_kernel()
    S.mi355_event_destroy(e0); S.mi355_event_destroy(e1); S.mi355_stream_destroy(st)
    t = ms.value / iters * 1e-3
    ops = 2.0 * n * c * k * k * hw * hw * batch
    return {"workload": "BASELINE config[1]: single 3x3 s1 conv 256->256 ch, 52x52, batch 32, uint8 in / int8 w (random uint8 operands), one kernel per launch",
            "us_per_launch": round(t * 1e6, 2), "achieved": round(ops / t / 1e12, 1), "unit": "TOP/s", "peak": round(PEAK_INT8_TOPS, 1),
            "frac": round(ops / t / 1e12 / PEAK_INT8_TOPS, 4), "ops_per_launch": ops, "launches": iters, "kernel_family": fam,
            "note": "HIP events on the launch stream around back-to-back launches, alone on the device, after the timed region; never part of `value`"}


def flush_c_stdio():
    """RCCL prints its version banner through C stdio; when stdout is a pipe that text would otherwise appear at process
    exit, after the JSON line."""
    import ctypes
    sys.stdout.flush()
    try:
        ctypes.CDLL(None).fflush(None)
    except OSError:
        pass


class StdoutGuard:
    """Everything any library writes to file descriptor 1 while the bench runs (RCCL's version banner, gloo's "[Gloo] Rank .." lines from
    C++ iostreams) goes to stderr; `emit` puts the ONE JSON line on the real stdout."""

    def __init__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)

    def emit(self, line):
        flush_c_stdio()
        os.dup2(self.saved, 1)
        print(line, flush=True)
        os.dup2(2, 1)


def self_launch(args):
    """`python bench.py --gpus N` with no launcher around it (WORLD_SIZE unset): re-run this command under torch.distributed.run,
    one rank per GPU on this node, rendezvous on 127.0.0.1 (the container's hostname may not resolve).  Rank 0 of the child job
    prints the ONE JSON line on the inherited stdout; the launcher's exit code is ours."""
    import subprocess
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC only on this host driver (RCCL across processes)
    env.setdefault("OMP_NUM_THREADS", "1")
    # --standalone: the launcher's own c10d rendezvous on a port IT binds (127.0.0.1:0) -- no bind-then-close-then-rebind window as with a
    # port picked here
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           os.path.abspath(__file__)] + sys.argv[1:]
    print("[bench] no launcher around --gpus %d: re-running as  %s" % (args.gpus, " ".join(cmd)), file=sys.stderr, flush=True)
    sys.exit(subprocess.call(cmd, env=env))


def dist_init(backend, rank, world, dev=None):
    """Rendezvous (env:// on 127.0.0.1) + the host-side group the timing barriers run on.  Returns (dist, timing group or None)."""
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(29500 + os.getppid() % 2000))  # the launcher sets it; this default only serves BENCH_FORCE_DIST
    os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")  # one node; the container's hostname may not resolve
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        # the barriers that bracket the timed region run on the host (gloo): an RCCL barrier is an all-reduce kernel + its launch and
        # wait, ~0.6 ms measured on one rank -- 10 % of a 20-step region -- and it says nothing a host barrier between ranks that have
        # each synchronised their device does not
        try:
            tgroup = dist.new_group(backend="gloo")
        except Exception as e:  # noqa: BLE001
            print(f"[bench] gloo group unavailable ({e}); timing barriers stay on RCCL", file=sys.stderr)
            tgroup = None
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
        tgroup = None
    return dist, tgroup


def bcast_packed(dist, torch, rank, packed, dev):
    """The one collective of the path: rank 0's packed quantized weights (network_export_packed bytes) to every rank.  `dev` = the
    rank's cuda device (RCCL over xGMI) or None (gloo, host tensors: the CPU dry run and tests).  Returns (uint8 tensor, ms)."""
    size_t = torch.tensor([packed.size if rank == 0 else 0], dtype=torch.int64, device=dev)
    dist.broadcast(size_t, 0)
    nbytes = int(size_t.item())
    blob = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    if rank == 0:
        blob.copy_(torch.from_numpy(packed))
    if dev is not None:
        torch.cuda.synchronize()
    t0 = time.time()
    dist.broadcast(blob, 0)
    if dev is not None:
        torch.cuda.synchronize()
    return blob, (time.time() - t0) * 1e3


def image_shard(rank, world, batch):
    """global image indices [start, stop) of this rank's batch: weak scaling, `batch` images per rank (sharding.shard_range)"""
    from yolo_quantization_amd.sharding import shard_range
    return shard_range(world * batch, rank, world)


def dry_dist(args, rank, world):
    """--dry-dist: the multi-rank start-up on CPU (gloo, host-only prep, no kernels).  Everything bench.py does across ranks runs: rendezvous,
    rank 0 reads + preps + packs, broadcast of the packed bytes, import on the other ranks, image sharding, barrier-bracketed timed region
    (empty steps) with the MAX over ranks.  Checked: every rank's re-exported packed state has rank 0's SHA-256; shards are disjoint and
    cover the global batch."""
    import hashlib
    import numpy as np
    import torch
    from yolo_quantization_amd import binding, synth
    guard = StdoutGuard()
    dist, tgroup = dist_init("gloo", rank, world)
    wts = f"/tmp/bench_dry_{os.getpid()}.weights"
    if rank == 0:
        synth.synth_weights(args.cfg, wts, seed=1234)
        net = binding.Net(args.cfg, wts, batch=args.batch)
        net.prepare_host_only(1.0 / 255.0, 0)
        packed = net.export_packed()
    else:
        net = binding.Net(args.cfg, None, batch=args.batch)
        packed = None
    blob, bcast_ms = bcast_packed(dist, torch, rank, packed, None)
    if rank != 0:
        net.import_packed_host(blob.numpy())
    digest = hashlib.sha256(net.export_packed().tobytes()).digest()
    mine = torch.tensor(list(digest) + list(image_shard(rank, world, args.batch)), dtype=torch.int64)
    everyone = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(everyone, mine)
    dist.barrier(group=tgroup) if tgroup is not None else dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        pass  # a step launches kernels; there is no GPU here
    dist.barrier(group=tgroup) if tgroup is not None else dist.barrier()
    t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    same = all(bool((e[:32] == everyone[0][:32]).all()) for e in everyone)
    spans = [(int(e[32]), int(e[33])) for e in everyone]
    covered = sorted(spans) == [(r * args.batch, (r + 1) * args.batch) for r in range(world)]
    net.close()
    if rank == 0:
        os.remove(wts)
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        guard.emit(json.dumps({"dry_dist": True, "n_gpus": world, "backend": "gloo", "packed_bytes": int(blob.numel()), "weight_broadcast_ms": round(bcast_ms, 3),
                          "packed_sha256_rank0": digest.hex(), "packed_state_identical_on_all_ranks": same, "image_shards": spans,
                          "shards_cover_global_batch": covered, "max_over_ranks_s": float(t.item()),
                          "note": "CPU dry run of the multi-rank start-up; no kernels ran, nothing here is a throughput figure"}))
    sys.exit(0 if (same and covered) else 1)


def main():
    args = parse_args()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if "WORLD_SIZE" not in os.environ and args.gpus > 1:
            self_launch(args)  # does not return
        print(f"[bench] --gpus {args.gpus} but the launcher started {world} ranks", file=sys.stderr)
        sys.exit(2)
    if args.dry_dist:
        dry_dist(args, rank, world)  # does not return
    guard = StdoutGuard()
    import numpy as np
    import torch  # first: its bundled HIP runtime must be the one the process uses (same SONAME as /opt/rocm's)
    if not torch.cuda.is_available():
        print("[bench] no GPU visible: the INT8 path has no CPU fallback", file=sys.stderr)
        sys.exit(3)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    force_dist = os.environ.get("BENCH_FORCE_DIST") == "1"  # exercise the RCCL start-up path on one rank
    tgroup = None
    if world > 1 or force_dist:
        dist, tgroup = dist_init("nccl", rank, world, dev)

    from yolo_quantization_amd import binding, synth
    binding.init(local_rank)
    if os.environ.get("BENCH_DEBUG_FLAGS"):  # A/B switches between equivalent kernel variants (mi355_debug_flags)
        binding.shim().mi355_debug_flags(int(os.environ["BENCH_DEBUG_FLAGS"]))
    if os.environ.get("BENCH_FORCE_TILE"):  # A/B: "bm,bn[,nt]" for every conv_rows launch (mi355_conv_set_tile)
        t = [int(v) for v in os.environ["BENCH_FORCE_TILE"].split(",")]
        binding.shim().mi355_conv_set_tile(t[0] | ((t[2] if len(t) > 2 else 0) << 16), t[1])
    for _ in range(int(os.environ.get("BENCH_DUMMY_STREAMS", "0"))):  # experiment: streams created before ours shift the stream -> hardware-queue mapping
        _s = ctypes.c_void_p()
        binding.shim().mi355_stream_create(ctypes.byref(_s))
    B = args.batch
    wts = f"/tmp/bench_yolov3_tiny_{os.getpid()}.weights"

    # ---- model: rank 0 reads the weights file, preps and packs; the packed bytes travel by RCCL broadcast
    if rank == 0:
        synth.synth_weights(args.cfg, wts, seed=1234, small_m_channels=args.small_m_channels)
        net = binding.Net(args.cfg, wts, batch=B, gpu=local_rank, use_graph=args.graph, keep_head_float=False)
        net.prepare_fixed(1.0 / 255.0, 0)
        packed = net.export_packed()
    else:
        net = binding.Net(args.cfg, None, batch=B, gpu=local_rank, use_graph=args.graph, keep_head_float=False)
        packed = None
    bcast_ms = 0.0
    if world > 1 or force_dist:
        blob, bcast_ms = bcast_packed(dist, torch, rank, packed, dev)
        nbytes = int(blob.numel())
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
    img0, img1 = image_shard(rank, world, B)  # this rank's images of the global batch (weak scaling: B per rank)
    x = synth.synth_image_u8(in_c, in_h, in_w, seed=1000 + img0 // max(B, 1), batch=img1 - img0)
    real = None
    if args.input == "realimg":  # (timing on real-image statistics: the same photograph in every slot of every instance's batch)
        g = np.load(os.path.join(ROOT, "tests", "golden", "realimg_416.npz"))
        real = np.ascontiguousarray(np.broadcast_to(g["input_u8"].reshape(1, in_c, in_h, in_w), (img1 - img0, in_c, in_h, in_w)))
        x = real
    net.push_input(x)
    net.sync()

    # ---- further batches in flight: instances built from the same packed bytes, each with its own batch of images
    nets = [net]
    for k in range(1, max(1, args.inflight)):
        # the fourth instance runs on the device's default stream: HIP maps every created stream onto three of the device's four
        # hardware queues and keeps the fourth for that one (a fourth created stream shares a queue: 0.272 -> 0.30 ms per step)
        nk = net.replica(default_stream=(k == 3 and not args.graph))
        nk.push_input(real if real is not None else synth.synth_image_u8(in_c, in_h, in_w, seed=1000 + rank + 7919 * k, batch=B))
        nk.sync()
        nets.append(nk)

    plan = 1 if len(nets) > 1 else 0  # network_replica switched parent and replicas to the throughput plan
    if os.environ.get("BENCH_PLAN"):  # A/B: 0 = latency plan (whole-chip kernels) although batches are in flight, 1 = throughput plan
        plan = int(os.environ["BENCH_PLAN"])
        for nk in nets:
            nk.set("plan", plan)
    plan_name = "throughput (kernels of which two workgroups share a CU)" if plan else "latency (every launch sized to fill the chip alone)"

    def barrier():
        # this rank's own work first (every instance's stream, then the whole device), THEN the cross-rank barrier and a last
        # device synchronise: the collective's kernel must not run beside the tail of the timed steps (it spins on CUs), and a
        # rank only enters the barrier when it is done -- the time after it is the slowest rank's
        for nk in nets:
            nk.sync()
        torch.cuda.synchronize()
        if world > 1 or force_dist:
            dist.barrier(group=tgroup) if tgroup is not None else dist.barrier()
            torch.cuda.synchronize()

    # (the determinism self-check used to run HERE, in front of the warm-up; it now runs after the timed region and its legs -- see below: its passes,
    # with their single-workgroup checksum kernels, leave the device in a state from which a 20-step region still measures 0.256-0.258 ms per step
    # after 165 plain steps, where the same region without them measures 0.246: profiles/r05_warmup_curve.log)
    # (round 6, ADVICE r05 / VERDICT r05 #5: the rounds 1-4 protocol -- W warm-up steps, then K timed steps, nothing in front of them -- once, BEFORE the
    # pre-roll, so that the line carries both figures of the same run: `no_preroll` is comparable with BENCH_r01-r04, `value` with r05 on)
    no_preroll = None
    if args.preroll > 0 and not args.no_preroll_leg:
        for i in range(args.warmup):
            nets[i % len(nets)].forward()
        barrier()
        t0 = time.perf_counter()
        for i in range(args.steps):
            nets[i % len(nets)].forward()
        barrier()
        np_dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([np_dt], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            np_dt = float(t.item())
        no_preroll = {"ms_per_step": round(np_dt / args.steps * 1e3, 4), "value": round(world * B * args.steps / np_dt, 1), "steps": args.steps, "warmup": args.warmup,
                      "note": "the same K steps timed the way rounds 1-4 timed them (W warm-up steps on a device that has just been set up, no pre-roll), "
                              "run BEFORE the pre-roll of this line's `value`; a short region on a device whose clocks are still settling (DESIGN.md 4.6)"}
    for i in range(max(args.preroll, 0)):  # untimed, unsynchronised: the same in-flight load as the timed steps (see --preroll)
        nets[i % len(nets)].forward()
    for i in range(args.warmup):
        nets[i % len(nets)].forward()
    barrier()
    # per-layer HIP events (launch stream) are recorded on every PROF_STRIDE-th step of the timed region.  An event costs
    # ~3.9 us of stream time: ~100 us for the 26 of a profiled step.  Measured in one box (400 steps): 0.377 ms / step with every
    # 8th step profiled, 0.365 with a single profiled step -- 3 % of `value` went into its own instrumentation.  Every 32nd step
    # (at least one) keeps that below 1 % on long runs (1.4 % at --steps 20).
    # A short timed region also sees the device's start-up: after any idle phase of more than ~2 ms the first ~25 steps run up
    # to 8 % slower (tools/step_curve.py: 0.400 0.388 0.376 0.370 0.368 .. ms per step in groups of five; --warmup 5
    # --steps 20 gives 0.38, --warmup 50 0.361, --warmup 200 0.355 in the same box).  Nothing is done about that here.
    prof_stride = max(1, int(os.environ.get("BENCH_PROF_STRIDE", "32")))
    ninfl = len(nets)

    def arm_events(nsteps_of_net0):
        """per-layer events on every prof_stride-th forward of nets[0], counted back from its LAST forward of the coming region (the
        first ones run on a device that is still ramping up); returns the number of profiled steps"""
        phase = (nsteps_of_net0 - 1) % prof_stride
        n = 0 if (args.graph or nsteps_of_net0 < 1) else min((nsteps_of_net0 - 1 - phase) // prof_stride + 1, 64)
        net.profile_begin(n, prof_stride, phase)
        return n

    prof_steps = arm_events((args.steps + ninfl - 1) // ninfl) if (ninfl == 1 or args.inflight_events) else arm_events(0)
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        nets[i % ninfl].forward()
    # (round 4, measured and not kept: one host thread per instance queues its steps -- host_issue_ms 1.2-1.9 -> 0.55-0.64 for 20 steps, the
    # region itself 0.285-0.291 -> 0.291-0.302 ms per step: what a 20-step region loses against a long run is the fill and drain of a
    # four-deep pipeline, not the host)
    t_issued = time.perf_counter() - t0  # host time to queue the K steps (reported as `host_issue_ms`: launch-bound if close to the total)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms_per_step = dt / args.steps * 1e3
    value = world * B * args.steps / dt
    flight_prof = net.profile_read() if (prof_steps and rank == 0) else None

    # ---- serial leg (only with several batches in flight): the same steps on ONE instance, nothing else on the device.  A kernel that
    # shares the CUs with another batch's kernels runs longer than alone, so per-kernel durations -- the roofline, the per-layer
    # table, and what rocprofv3's kernel trace of `--inflight 1` shows -- are taken here; never part of `value`.
    serial = None
    ser_dt = dt
    ser_steps = args.steps
    if ninfl > 1:
        ser_steps = max(args.serial_steps, 2)
        prof_steps = arm_events(ser_steps)
        barrier()
        t0 = time.perf_counter()
        for _ in range(ser_steps):
            net.forward()
        barrier()
        ser_dt = time.perf_counter() - t0
        serial = {"ms_per_step": round(ser_dt / ser_steps * 1e3, 4), "value": round(B * ser_steps / ser_dt, 1), "steps": ser_steps,
                  "note": "one batch at a time on one stream (per GPU), run right after the timed region; per-kernel figures (roofline, --layers) come from here"}

    serial_prof = net.profile_read() if (prof_steps and rank == 0) else None
    rows_layers = [i for i, inf in enumerate(net.info) if inf["type"] == binding.T_CONV and inf["c"] % 64 == 0 and net.conv_kernel(i) == 5]

    # ---- latency-plan leg (rank 0, only when the timed region ran the throughput plan): the same net one batch at a time with the
    # whole-chip kernels of rounds 1-2 (128 x 384 row-image tiles, conv_ws3), so that their roofline stays on record next to
    # the half-CU kernels'.  Never part of `value`.
    latency_leg = None
    if ninfl > 1 and plan == 1 and rank == 0 and world == 1 and not args.graph and not args.no_extra_legs:
        net.set("plan", 0)
        for _ in range(4):
            net.forward()
        n_lat = arm_events(32)
        net.sync()
        t0 = time.perf_counter()
        for _ in range(32):
            net.forward()
        net.sync()
        lat_dt = time.perf_counter() - t0
        lat_prof = net.profile_read()
        lat_rows = [i for i, inf in enumerate(net.info) if inf["type"] == binding.T_CONV and inf["c"] % 64 == 0 and net.conv_kernel(i) == 5]
        net.set("plan", 1)
        latency_leg = (n_lat, lat_prof, lat_dt, lat_rows)

    # ---- sustained rate of the dominant kernel (rank 0, batches in flight only): every row-image launch of the step repeated on all
    # instances at once (the layer range knob of the host: forward_network_gpu runs that one layer on the tensors the last pass left),
    # wall time / number of launches = what one launch costs the chip when the chip is kept full of this kernel -- no launch gap, fill or
    # tail between dependent launches, which is how the kernel runs in the timed region.  (tools/layer_flood.py does this for every layer.)
    sustained = sustained33 = energy = None
    extra_legs = ninfl > 1 and rank == 0 and world == 1 and not args.graph and not args.no_extra_legs  # (world > 1: every other rank would sit in the final barrier meanwhile)
    s33_layers = [i for i, inf in enumerate(net.info) if inf["type"] == binding.T_CONV and inf["size"] == 3 and inf["stride"] == 1 and inf["c"] > 3]
    if extra_legs and (rows_layers or s33_layers):
        per = {}

        def flood(lo, hi):
            for nk in nets:
                nk.set("range_lo", lo); nk.set("range_hi", hi)
            for _ in range(20):
                for nk in nets:
                    nk.forward()
            for nk in nets:
                nk.sync()
            # (round 6: 200 launches per instance instead of 40 -- a 3 ms burst right after a synchronisation still sees the device settle its
            # clocks: the same layers read 3-5 % longer at 60 launches than at 1 000, profiles/r06_flood_reps_60_vs_1000.log)
            reps = 200
            t0 = time.perf_counter()
            for _ in range(reps):
                for nk in nets:
                    nk.forward()
            for nk in nets:
                nk.sync()
            return (time.perf_counter() - t0) / (reps * ninfl) * 1e6
        for i in sorted(set(rows_layers) | set(s33_layers)):
            # a conv + maxpool launch is one launch: the pool layer belongs to the range when the conv's kernel runs it.  plan_fusion marks
            # CANDIDATES; the launcher has the last word (under the throughput plan the 128- / 256-channel layers run the row-image kernel,
            # which has no fused pool: the host then clears the flag and the pool is a launch of its own) -- so the flag is read again after
            # the warm-up passes, and a stand-alone pool is timed apart (round 5 counted it into the conv's launch)
            cand = net.fuses_next(i) and net.info[i + 1]["type"] == binding.T_MAXPOOL
            us = flood(i, i + 2 if cand else i + 1)
            fused = cand and net.fuses_next(i)
            us_with_pool = None
            if cand and not fused:
                us_with_pool = us
                us = flood(i, i + 1)
            ops = conv_layer_work(net.info[i], B)[0]
            per[i] = {"layer": i, "conv": "%d->%d @%d%s" % (net.info[i]["c"], net.info[i]["n"], net.info[i]["out_h"], " + maxpool" if fused else ""),
                      "us_per_launch": round(us, 2), "tops": round(ops / us / 1e6, 1), "frac": round(ops / us / 1e6 / PEAK_INT8_TOPS, 4), "ops": ops}
            if us_with_pool is not None:
                per[i]["us_with_its_standalone_maxpool_launch"] = round(us_with_pool, 2)  # (what BENCH_r05's figure for this layer contained)
        for nk in nets:
            nk.set("range_lo", 0); nk.set("range_hi", 0)

        def agg(which, note):
            rows = [per[i] for i in which if i in per]
            if not rows:
                return None
            o, u = sum(r["ops"] for r in rows), sum(r["us_per_launch"] for r in rows)
            u5 = sum(r.get("us_with_its_standalone_maxpool_launch", r["us_per_launch"]) for r in rows)
            return {"achieved": round(o / u / 1e6, 1), "frac": round(o / u / 1e6 / PEAK_INT8_TOPS, 4), "us": round(u, 2),
                    "frac_with_standalone_maxpool_launches": round(o / u5 / 1e6 / PEAK_INT8_TOPS, 4), "us_with_standalone_maxpool_launches": round(u5, 2),
                    "layer_set": "L" + ", L".join(str(r["layer"]) for r in rows),
                    "launches": [{k: v for k, v in r.items() if k != "ops"} for r in rows], "note": note}
        sustained = agg(rows_layers, f"the row-image launches of the step ONLY (not the north-star layer set: see conv3x3_s1_aggregate.sustained), each repeated 200 times on all "
                                     f"{ninfl} instances at once; host wall time / launches (no events: the launches overlap).  `frac` above is the strict per-launch figure "
                                     "(one launch alone on the device, launch gap, fill and tail included); this is the rate the kernel sustains when the chip is kept full of it, "
                                     "as in the timed region")
        sustained33 = agg(s33_layers, f"EVERY 3x3 stride-1 conv with c > 3 (the north-star target's layer set), whichever kernel serves it, its FUSED maxpool included: each launch "
                                      f"repeated 200 times on all {ninfl} instances at once, host wall time / launches.  A maxpool that runs as a launch of its own behind the conv "
                                      "(L8 / L10 under the throughput plan: the row-image kernel has no fused pool) is a byte kernel, not part of the conv: timed apart "
                                      "(`us_with_its_standalone_maxpool_launch`; `frac_with_standalone_maxpool_launches` is the aggregate the way BENCH_r05 counted it)")

    # ---- energy leg (rank 0, one GPU): the in-flight step repeated for >= 0.3 s with the device's hwmon power / clock files sampled on a host
    # thread (every 5 ms, second half of the samples: the sensor averages over a window).  Never part of `value`.
    if extra_legs:
        from yolo_quantization_amd.hwmon import Sampler
        smp = Sampler(local_rank)
        if smp.available():
            e_steps = max(400, int(0.3 / max(ms_per_step * 1e-3, 1e-6)))

            def region():
                for nk in nets:
                    nk.sync()
                t0 = time.perf_counter()
                for i in range(e_steps):
                    nets[i % ninfl].forward()
                for nk in nets:
                    nk.sync()
                return time.perf_counter() - t0
            region()
            e_dt, e_w, e_mhz, e_n = smp.run(region)
            if e_w is not None:
                energy = {"joules_per_image": round(e_w * e_dt / (e_steps * B), 6), "mean_socket_watts": round(e_w, 1), "mean_shader_mhz": round(e_mhz, 0),
                          "ms_per_step": round(e_dt / e_steps * 1e3, 4), "steps": e_steps, "samples": e_n,
                          "note": f"{e_steps} in-flight steps after the timed region, amdgpu hwmon power1_input / freq1_input every 5 ms (second half of the samples); "
                                  "the chip's cap is 1 400 W: at the cap a step costs its energy, not its instruction count (DESIGN.md 4.4)"}

    # ---- BASELINE config[1] microbench leg (rank 0, one GPU): the one-kernel MFMA-I8 shape with its own roofline fraction
    microbench = None
    if extra_legs and os.path.basename(args.cfg) == "yolov3-tiny_quant.cfg":
        try:
            microbench = microbench_config1(binding)
        except Exception as e:  # noqa: BLE001
            print(f"[bench] config[1] microbench leg failed ({e})", file=sys.stderr)

    # ---- determinism self-check (a race detector for kernels scheduled by hand: counted waits, registers reloaded in place): N passes over the
    # resident input on every instance at once (their passes overlap on the device like the timed steps), a device-side order-independent checksum
    # of the yolo outputs after each pass; a pass that differs from the first fails the run.  After the timed region and its legs, never inside them.
    selfcheck = None
    if args.selfcheck_passes > 0 and not args.graph:
        for nk in nets:
            nk.selfcheck(args.selfcheck_passes)
        bad = sum(nk.selfcheck_result() for nk in nets)
        selfcheck = {"passes": args.selfcheck_passes, "instances": ninfl, "passes_differing_from_the_first": bad}
        if bad:
            raise SystemExit(f"bench.py: determinism self-check FAILED: {bad} of {ninfl} x {args.selfcheck_passes} passes over the same input gave other yolo outputs")

    def layer_table(nprof, ms, region_dt, region_steps):
        """per-layer ms from the event intervals of `nprof` profiled steps (nets[0]); region_dt / region_steps calibrate the cost of an
        event when the region ran this instance alone (None: several batches were in flight, only the empty-interval cost is known)"""
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
        if region_dt is not None and region_steps > nprof > 0:
            t_prof = float(sum(ms)) / nprof                       # ms of one profiled step, events included
            t_unprof = (region_dt * 1e3 - nprof * t_prof) / (region_steps - nprof)
            ev_cost = min(ev_cost, max((t_prof - t_unprof) / len(ms), 0.0))
        rows = []
        for i, inf in enumerate(net.info):
            t_ms = max(float(ms[i + 1]) / nprof - ev_cost, 1e-6)
            row = {"i": i, "type": inf["type"], "ms": round(t_ms, 5)}
            rd_b, wr_b = launch_bytes(net, i, B)
            launches = not ((i > 0 and fused[i - 1]) or (i > 0 and net.fuses_next(i - 1)) or inf["type"] in (binding.T_ROUTE, binding.T_YOLO))
            if inf["type"] == binding.T_CONV:
                ops, _ = conv_layer_work(inf, B)
                # gbs: algorithmic bytes of the launch that runs (fused launches: input + weights + the tensors it STORES)
                row.update(tops=round(ops / (t_ms * 1e-3) / 1e12, 2), gbs=round((rd_b + wr_b) / (t_ms * 1e-3) / 1e9, 1), bytes=rd_b + wr_b,
                           fused_next=bool(net.fuses_next(i)), k=inf["size"], c=inf["c"], n=inf["n"], hw=inf["out_h"], ops=ops,
                           kernel_family=int(net.conv_kernel(i)))  # mi355_last_conv_kernel's code of the launch that served the layer
            elif launches and t_ms > 1e-3:
                row.update(gbs=round((rd_b + wr_b) / (t_ms * 1e-3) / 1e9, 1), bytes=rd_b + wr_b)
            rows.append(row)
        return rows, ev_cost, max(float(ms[0]) / nprof - ev_cost, 0.0)

    def on_rows_kernel(i, inf, rows=None):
        """conv_rows16_i8_kernel / conv_rows_i8_kernel launches: 64-byte channel chunks, served by the implicit-GEMM family (the host records
        which kernel family took each conv of the last step: conv_small / conv1x1 / conv_ws3 take the others)."""
        return i in (rows_layers if rows is None else rows)

    def rows_rate(rows, which=None):
        ops = sum(r["ops"] for i, r in enumerate(rows) if on_rows_kernel(i, net.info[i], which))
        ms_ = sum(r["ms"] for i, r in enumerate(rows) if on_rows_kernel(i, net.info[i], which))
        return ops, ms_

    def s33_rate(rows):
        sel = [r for i, r in enumerate(rows) if net.info[i]["type"] == binding.T_CONV and net.info[i]["size"] == 3
               and net.info[i]["stride"] == 1 and net.info[i]["c"] > 3]
        return sum(r["ops"] for r in sel), sum(r["ms"] for r in sel)

    # ---- roofline of the dominant kernel (the MFMA implicit-GEMM conv), from HIP events on the launch stream
    roof = None
    layers = []
    if prof_steps and rank == 0:
        nprof, ms = serial_prof
        layers, ev_cost, in_layout_ms = layer_table(nprof, ms, ser_dt, ser_steps)
        mf_ops, mf_ms = rows_rate(layers)
        s33_ops, s33_ms = s33_rate(layers)
        all_ops = sum(r.get("ops", 0.0) for r in layers)
        all_ms = sum(r["ms"] for r in layers)
        nlaunch = sum(1 for i, inf in enumerate(net.info) if on_rows_kernel(i, inf))
        nconv = sum(1 for inf in net.info if inf["type"] == binding.T_CONV)
        achieved = mf_ops / (mf_ms * 1e-3) / 1e12 if mf_ms else 0.0
        (traffic, traffic_rows), traffic_src = pmc_traffic_per_launch() if pmc_traffic_per_launch()[0] else ((None, None), None)
        # Headline (round 6): the NORTH-STAR layer set -- every 3x3 stride-1 conv with c > 3, whichever kernel serves it, fused maxpools
        # included -- per launch alone on the device.  Rounds 1-5 put the row-image launches (the five best) here; they are `row_image_launches` now.
        n33 = sum(1 for inf in net.info if inf["type"] == binding.T_CONV and inf["size"] == 3 and inf["stride"] == 1 and inf["c"] > 3)
        ach33 = s33_ops / (s33_ms * 1e-3) / 1e12 if s33_ms else 0.0
        roof = {"bound": "mfma", "kernel": f"the {n33} 3x3 stride-1 conv launches with c > 3 (BASELINE north_star's layer set: conv_pool16 / conv_small_pool / conv_mid_pool / conv_rows / conv_rows16 on "
                          f"V_MFMA_I32_*_I8, fused maxpools included: {100 * s33_ms / all_ms:.0f}% of the step's time and {100 * s33_ops / all_ops:.0f}% of its operations)",
                "achieved": round(ach33, 2), "peak": round(PEAK_INT8_TOPS, 1), "unit": "TOP/s",
                "frac": round(ach33 / PEAK_INT8_TOPS, 4),
                "row_image_launches": {"kernel": f"conv_rows16_i8_kernel / conv_rows_i8_kernel (row-image MFMA implicit GEMM on 64-channel chunks: {nlaunch} of the step's {nconv} conv launches, "
                                                 f"{100 * mf_ms / all_ms:.0f}% of its time and {100 * mf_ops / all_ops:.0f}% of its operations) -- what `roofline.frac` covered in rounds 1-5",
                                       "achieved": round(achieved, 2), "frac": round(achieved / PEAK_INT8_TOPS, 4),
                                       "ops_per_launch_avg": mf_ops / max(nlaunch, 1), "ms_per_launch_avg": round(mf_ms / max(nlaunch, 1), 5),
                                       "traffic": traffic_rows if os.path.basename(args.cfg) == "yolov3-tiny_quant.cfg" else None},
                "measured_on": ("the serial leg after the timed region (one batch at a time: the kernel alone on the device, as in rocprofv3's trace of --inflight 1)"
                                if ninfl > 1 else "the timed region"),
                "traffic": traffic if os.path.basename(args.cfg) == "yolov3-tiny_quant.cfg" else None,
                "traffic_unit": "HBM bytes per launch (rocprofv3 PMC: FETCH_SIZE x2 + WRITE_SIZE, separate passes)",
                "traffic_source": f"committed profile {traffic_src} -- not measured in this run" if traffic_src else None,
                "ops_per_launch_avg": s33_ops / max(n33, 1), "ms_per_launch_avg": round(s33_ms / max(n33, 1), 5),
                "conv3x3_s1_aggregate": {"tops": round(s33_ops / (s33_ms * 1e-3) / 1e12, 1) if s33_ms else None,
                                         "frac": round(s33_ops / (s33_ms * 1e-3) / 1e12 / PEAK_INT8_TOPS, 4) if s33_ms else None,
                                         "ms": round(s33_ms, 5), "layers": "every 3x3 stride-1 conv with c > 3"},
                # what a loop of NOTHING but V_MFMA_I32_32X32X32_I8 sustains on this chip depends on the operand bytes (power
                # management lowers the shader clock): 4 760-4 940 TOP/s on zeros, 3 490 on uniform random bytes
                # (tools/ubench/mfma_data_power.hip, profiles/r02_v3_ubench_mfma_data_power.log).  `peak` / `frac` stay nominal.
                "mfma_only_loop_random_operands": {"tops": 3490.0, "frac_of_it": round(ach33 / 3490.0, 4),
                                                   "source": "profiles/r02_v3_ubench_mfma_data_power.log -- not measured in this run"},
                "input_layout_ms": round(in_layout_ms, 5),
                "event_overhead_ms": round(ev_cost, 5)}
        # ---- HBM rows ("GB/s for L0/L2/pools"): the few-channel convs with their fused pools and every stand-alone glue launch.
        # algorithmic = launch_bytes() (fused launches count what they read and STORE); counter = the same launch's bytes in the newest
        # committed PMC passes (FETCH_SIZE x2 + WRITE_SIZE, fabric side: Infinity-Cache hits included) over THIS run's duration.
        pmc_rows, pmc_src = pmc_kernel_rows()

        def pmc_bytes_of(i, inf):
            code = net.conv_kernel(i) if inf["type"] == binding.T_CONV else -1
            want = {1: "conv_first_mfma", 2: ("conv_mid_pool" if inf["c"] == 64 else f"conv_small_pool_kernel<{inf['c']},")}.get(code)
            if inf["type"] == binding.T_MAXPOOL:
                want = "maxpool_u8_kernel"
            for r in pmc_rows:
                if want and want in r["kernel"] and (inf["type"] != binding.T_MAXPOOL or r["grid_threads"] == B * inf["outputs"] // 16):
                    return (r["hbm_read_bytes_per_launch"] or 0) + (r["hbm_write_bytes_per_launch"] or 0)
            return None

        hbm_rows = []
        if os.path.basename(args.cfg) == "yolov3-tiny_quant.cfg":
            for r in layers:
                inf = net.info[r["i"]]
                if "bytes" not in r or (inf["type"] == binding.T_CONV and not (inf["c"] < 64 and inf["size"] == 3)):
                    continue
                cb = pmc_bytes_of(r["i"], inf)
                hbm_rows.append({"layer": r["i"], "what": ("conv %d->%d 3x3%s" % (inf["c"], inf["n"], " + maxpool" if r.get("fused_next") else "")) if inf["type"] == binding.T_CONV else "maxpool",
                                 "us": round(r["ms"] * 1e3, 2), "algorithmic_bytes": r["bytes"], "gbs": r["gbs"], "frac": round(r["gbs"] / PEAK_HBM_GBS, 4),
                                 "counter_bytes": cb, "counter_gbs": round(cb / (r["ms"] * 1e-3) / 1e9, 1) if cb else None,
                                 "counter_frac": round(cb / (r["ms"] * 1e-3) / 1e9 / PEAK_HBM_GBS, 4) if cb else None})
        roof["hbm_layers"] = {"peak": PEAK_HBM_GBS, "unit": "GB/s", "rows": hbm_rows,
                              "note": "HBM-side rows: algorithmic bytes = input + weights + the tensors the (fused) launch stores; counter bytes from the committed PMC "
                                      f"passes ({pmc_src}; FETCH_SIZE x2 + WRITE_SIZE, not measured in this run) over this run's launch duration"}
        if flight_prof and ninfl > 1:  # the same kernels while other batches' kernels share the CUs (timed region)
            f_rows, _, _ = layer_table(flight_prof[0], flight_prof[1], None, 0)
            f_ops, f_ms = rows_rate(f_rows)
            roof["in_flight"] = {"achieved": round(f_ops / (f_ms * 1e-3) / 1e12, 2) if f_ms else None,
                                 "frac": round(f_ops / (f_ms * 1e-3) / 1e12 / PEAK_INT8_TOPS, 4) if f_ms else None,
                                 "ms_per_launch_avg": round(f_ms / max(nlaunch, 1), 5),
                                 "note": f"event intervals of the same launches inside the timed region, {ninfl} batches in flight: the kernel shares the "
                                         "CUs with other batches' kernels, its duration is no longer a measure of the kernel"}
        if sustained:
            roof["sustained"] = sustained
        if sustained33:
            roof["conv3x3_s1_aggregate"]["sustained"] = sustained33
        c33 = roof["conv3x3_s1_aggregate"]
        roof["north_star_target"] = ("0.40 of the dense INT8 MFMA peak wanted on the 3x3 stride-1 convs; measured over "
                                     + (sustained33["layer_set"] if sustained33 else "those layers") + ": "
                                     + (f"{sustained33['frac']} sustained (chip kept full of each launch), " if sustained33 else "")
                                     + f"{c33['frac']} per launch alone on the device")
        if energy:
            roof["energy"] = energy
        if microbench:
            roof["microbench_config1"] = microbench
        if latency_leg and latency_leg[0]:
            n_lat, lat_prof, lat_dt, lat_rows = latency_leg
            l_rows, _, _ = layer_table(lat_prof[0], lat_prof[1], lat_dt, 32)
            l_ops, l_ms = rows_rate(l_rows, lat_rows)
            l33_ops, l33_ms = s33_rate(l_rows)
            roof["latency_plan"] = {"kernel": "conv_rows16_i8_kernel<128, 384, ...> / conv_rows_i8_kernel<128, 384, ...> (one whole-CU workgroup per CU: the kernels of rounds 1-2), "
                                              f"{len(lat_rows)} launches per step",
                                    "achieved": round(l_ops / (l_ms * 1e-3) / 1e12, 2) if l_ms else None,
                                    "frac": round(l_ops / (l_ms * 1e-3) / 1e12 / PEAK_INT8_TOPS, 4) if l_ms else None,
                                    "ms_per_launch_avg": round(l_ms / max(len(lat_rows), 1), 5), "ms_per_step": round(lat_dt / 32 * 1e3, 4),
                                    "conv3x3_s1_aggregate_frac": round(l33_ops / (l33_ms * 1e-3) / 1e12 / PEAK_INT8_TOPS, 4) if l33_ms else None,
                                    "note": "the same net one batch at a time under MI355_PLAN_LATENCY, 32 steps after the serial leg; not the kernels that produced `value`"}
            for r in l_rows:
                r.pop("ops", None)
        for r in layers:
            r.pop("ops", None)
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
        if not args.no_cpu_omp and cpu and cpu["kind"] == "reference":
            # the reference's own all-cores build (Makefile MULTI_CORE=1: `#pragma omp parallel for` over the GEMM's output rows,
            # ref src/gemm.c:291) on every hardware thread of this host, bounded sample, after the timed region
            # (its loop is `omp parallel for` over M = the layer's filters, 16 .. 1024: on a 256-thread host every thread count past a few dozen
            # only adds fork / join cost -- 8, 32 and all threads are timed, the best is reported with its count)
            try:
                tried = []
                for nt in sorted({8, 32, os.cpu_count() or 1}):
                    if nt > (os.cpu_count() or 1):
                        continue
                    r = cpu_baseline(args.cfg, wts, args.cpu_omp_images, omp=True, threads=nt)
                    tried.append(r)
                cpu_omp = max(tried, key=lambda r: r["value"])
                cpu_omp["thread_counts_tried"] = {str(r["cores"]): round(r["value"], 3) for r in tried}
            except Exception as e:  # noqa: BLE001
                print(f"[bench] all-cores CPU baseline unavailable ({e})", file=sys.stderr)

    if rank == 0:
        out = {"metric": METRICS.get(os.path.basename(args.cfg), f"images/sec {os.path.basename(args.cfg)} INT8"), "value": round(value, 1), "unit": "images/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "warmup_passes_effective": args.warmup + max(args.preroll, 0) + (no_preroll["steps"] + no_preroll["warmup"] if no_preroll else 0),  # the pre-roll steps are queued right in front of the warmup steps (and the no_preroll leg ran before them)
               "no_preroll": no_preroll,
               "preroll_steps": max(args.preroll, 0),  # untimed steps of the same load that let the device's clocks settle (--preroll; the timed region is exactly `steps` steps)
               "ms_per_step": round(ms_per_step, 4),
               "host_issue_ms": round(t_issued * 1e3, 3),  # host time to queue the K steps of the timed region (its total is ms_per_step * steps)
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8 x s8 -> int32 (f64 or exact-integer requant, bit-identical)",
               "data": "synthetic" + ("" if args.input == "synthetic" else " weights, real-image fixture (tests/golden/realimg_416.npz) in every batch slot")
                       + (f", unfriendly model: {args.small_m_channels} filters per conv with multipliers ~2e-5" if args.small_m_channels else ""),
               "config": {"workload": "yolov3-tiny full net (cfg/yolov3-tiny_quant.cfg, leaky, per-channel quant), "
                                      f"batch {B}/GPU synthetic uint8 {in_h}x{in_w}, inputs resident in HBM (NCHW uint8)"
                                      if os.path.basename(args.cfg) == "yolov3-tiny_quant.cfg" else
                                      f"{os.path.basename(args.cfg)}, batch {B}/GPU synthetic uint8 {in_h}x{in_w}, inputs resident in HBM (NCHW uint8)",
                          "global_batch": world * B, "parallelism": f"image-sharded x{world}, RCCL weight broadcast once",
                          "batches_in_flight": ninfl, "kernel_plan": plan_name,
                          "launch": (f"{ninfl} batches in flight per GPU: step i runs on network instance i % {ninfl} (network_replica: own activations, input and "
                                     "HIP stream -- the fourth on the default stream, i.e. the fourth hardware queue; one copy of the packed weights), so the device overlaps the kernels of consecutive steps; every step is one "
                                     f"forward pass over its own batch of {B} images; " if ninfl > 1 else "one batch at a time on one stream; ")
                                    + ("hipGraph replay" if args.graph else
                                       (f"eager launches, per-layer HIP events on every {prof_stride}th forward of instance 0" if (ninfl == 1 or args.inflight_events)
                                        else "eager launches, no events inside the timed region (per-kernel figures come from the serial leg after it)")),
                          "weight_broadcast_ms": round(bcast_ms, 3)},
               "roofline": roof, "serial": serial, "cpu_baseline": cpu if world == 1 else "see the n_gpus=1 line (timed on rank 0 at N=1 only)",
               "accum_ref_f32_mode": ref_f32, "selfcheck": selfcheck}
        if cpu_omp:
            out["cpu_baseline_omp"] = cpu_omp
        if layers:
            os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
            json.dump({"ms_per_step": serial["ms_per_step"] if serial else ms_per_step, "layers": layers},
                      open(os.path.join(ROOT, "gpurun_out", f"bench_layers_n{world}.json" if os.path.basename(args.cfg) == "yolov3-tiny_quant.cfg"
                                        else f"bench_layers_{os.path.splitext(os.path.basename(args.cfg))[0]}_n{world}.json"), "w"), indent=1)
    try:
        os.remove(wts)
    except OSError:
        pass
    for nk in reversed(nets):  # replicas before their parent
        nk.close()
    if world > 1 or force_dist:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:  # the ONE JSON line, alone on stdout
        guard.emit(json.dumps(out))


if __name__ == "__main__":
    main()
