#!/usr/bin/env python3
"""Benchmark of the N-channel DDC ingest (BASELINE.json metric).

One step = one pass of the RX hot path (NCO mix, CIC /512, CIC compensator /2,
Hilbert / delay) over one batch of packed 12-bit ADC rows that already sit in
HBM.  Workload per GPU: the per-GPU shard of BASELINE config 5
(1024 tuned channels x 2^20 ADC samples each); with N GPUs the job is
1024*N channels (weak scaling) and every step ends with an RCCL all-gather of
the decimated SPEC+VOICE I/Q, as that config prescribes.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python bench.py --gpus 4            # starts 4 ranks itself, one per GPU
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Rank 0 prints ONE JSON line.  `--dry-run-gloo` rehearses the multi-rank plumbing
(launch, rendezvous, double-buffered all-gather, JSON) on CPUs with a pattern in
place of the kernels; its numbers mean nothing.
"""
import argparse
import hashlib
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "ua3reo-ddc-transceiver_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

METRIC = "ADC MSamples/s ingested per node, N-channel DDC; % HBM roofline"
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak
BYTES_PER_CHANNEL_SAMPLE = 1.5 + 8.0 / 1024.0  # packed ADC read + SPEC/VOICE int16 I/Q written
FS_ADC = 49152000
N_SIMD = 256 * 4               # MI355X: 256 CUs x 4 SIMDs
# the ingest kernel of the default call plan and the functions it shares with the other kernels
FRONT_SOURCES = [os.path.join(ROOT, "ua3reo-ddc-transceiver_amd", "csrc", f)
                 for f in ("ddc_rx_kernels.hip.h", "ddc_rx_mx_kernel.hip.h")]


class _StdoutToStderr:
    """Points file descriptor 1 at stderr for a while: stdout is reserved for the one JSON line."""

    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self.saved, 1)
        os.close(self.saved)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--channels", type=int, default=1024, help="channels per GPU")
    ap.add_argument("--samples", type=int, default=1 << 20, help="ADC samples per channel per step")
    ap.add_argument("--workload", choices=["ddc", "ddc+fft", "tx"], default="ddc",
                    help="ddc: RX chain (the headline metric); ddc+fft: RX chain + one 512-point panadapter "
                         "row per 512 SPEC samples (config 3 shape: --channels 256 --samples 4194304); "
                         "tx: DUC, --samples = DAC samples per channel (config 4 shape: --channels 256 "
                         "--samples 4194304)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-check", action="store_true")
    ap.add_argument("--no-copy-rate", action="store_true", help="skip the device-to-device copy rate in the roofline block")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the config-3 (ddc+fft) and config-4 (tx) shapes that the default run times after the headline")
    ap.add_argument("--no-settle", action="store_true",
                    help="time the --steps calls directly behind the warm-up steps (no untimed calls until the clock has settled)")
    ap.add_argument("--shared-stream", action="store_true",
                    help="one ADC stream fanned out to every channel (row pitch 0) instead of one stream per channel; "
                         "reported apart from the headline: the ADC bytes then come from L2, not HBM")
    ap.add_argument("--rx-plan", choices=["auto", "split", "valu"], default="auto",
                    help="auto: the library picks the RX call plan by size (two kernels for the bench shapes, the ingest "
                         "kernel with the window moments on the matrix cores); valu: the same plan with the vector-ALU "
                         "ingest kernel; split: the six separate kernels; for A/B timing on one box")
    ap.add_argument("--tune-all-hz", type=int, default=None,
                    help="experiment: tune every channel of the ddc workload to this frequency instead of 1 MHz + 48 kHz * channel")
    ap.add_argument("--force-gather", action="store_true",
                    help="run the per-step all-gather even with one rank (rehearses the N > 1 path on one GPU)")
    ap.add_argument("--gather", choices=["library", "torch"], default="library",
                    help="library: ddc_allgather_decimated_dev (RCCL behind the C ABI; ranks fall back to torch together, "
                         "and say so in the line, if its communicator cannot be built); torch: torch.distributed's "
                         "all_gather_into_tensor with the same double buffering")
    ap.add_argument("--dry-run-gloo", action="store_true",
                    help="no GPU: gloo ranks on CPUs, a pattern instead of the kernels; checks launch + gather + JSON")
    ap.add_argument("--master-port", type=int, default=0, help="rendezvous port when bench.py starts the ranks itself")
    return ap.parse_args(argv)


def launch_ranks(args, argv):
    """--gpus N without a launcher: start N ranks of this script (fresh processes; this
    parent has not touched the GPU and only waits), pass rank 0's JSON line through and
    fail if any rank fails."""
    import socket
    port = args.master_port
    if not port:
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out0 = procs[0].communicate()[0].decode()
    rcs = [procs[0].returncode] + [p.wait() for p in procs[1:]]
    sys.stdout.write(out0)
    sys.stdout.flush()
    bad = [(r, rc) for r, rc in enumerate(rcs) if rc != 0]
    if bad:
        sys.stderr.write("bench: ranks failed: %s\n" % bad)
        return 1
    return 0


def source_hash():
    h = hashlib.sha256()
    for path in FRONT_SOURCES:
        with open(path, "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def load_profile(name, nch, samples):
    """A committed PMC summary (profiles/<name>) if it was taken on this workload shape AND on
    the ingest kernel source that is in the tree now; rocprofv3 counters cannot be read from
    inside the run, so a summary of an older kernel is not replayed."""
    try:
        with open(os.path.join(ROOT, "profiles", name)) as f:
            t = json.load(f)
        if t["channels"] == nch and t["samples"] == samples and t.get("kernel_source_sha256_16") == source_hash():
            return t
    except (OSError, KeyError, ValueError):
        pass
    return None


def cpu_baseline(threads):
    """The CPU oracle (C restatement of the reference datapath, built here with
    gcc -O3 -march=native -fopenmp) on a bounded sample of the same workload: the
    window form of mixer + rx_cic with sixteen windows in the CPU's vector lanes
    (oracle/ddc_oracle.c, orc_fast_rx_front_run) and the 96 / 48 kHz stages of the
    scalar oracle, one channel per OpenMP task.  It has to reproduce the clock-level
    model bit for bit before it is timed.  Best of 5 after one warm-up."""
    import tempfile
    import numpy as np
    import ua3reo_ddc
    from oracle import oracle
    L = oracle.build_native_variant(tempfile.mkdtemp(prefix="ddc_oracle_"))
    samples, distinct, per_thread = 1 << 22, 4, 24
    rows = np.stack([ua3reo_ddc.pack_adc12(ua3reo_ddc.synth_adc(c, samples)) for c in range(distinct)])
    rb = rows.shape[1]
    n_ch = threads * per_thread
    phis = [ua3reo_ddc.phrase_from_frequency(1000000 + 48000 * c)[0] for c in range(n_ch)]
    a = oracle.rx_chain_packed_mt(L, rows, rb, distinct, distinct, 1 << 17, phis[:distinct], threads)
    b = oracle.rx_chain_packed(rows, rb, distinct, 1 << 17, phis[:distinct])
    if not (np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])):
        raise SystemExit("bench: the vectorised CPU chain disagrees with the clock-level oracle")
    t0 = time.perf_counter()
    oracle.rx_chain_packed(rows, rb, 1, samples, phis[:1], L=L)
    clock_one = samples / (time.perf_counter() - t0) / 1e6
    best, used = 0.0, threads
    for rep in range(6):
        t0 = time.perf_counter()
        _, _, used = oracle.rx_chain_packed_mt(L, rows, rb, distinct, n_ch, samples, phis, threads)
        dt = time.perf_counter() - t0
        if rep:
            best = max(best, n_ch * samples / dt / 1e6)
    one = 0.0
    for rep in range(3):
        t0 = time.perf_counter()
        oracle.rx_chain_packed_mt(L, rows, rb, distinct, 8, samples, phis[:8], 1)
        one = max(one, 8 * samples / (time.perf_counter() - t0) / 1e6)
    return best, one, used, (
        "best of 5: %d OpenMP threads x %d channel-passes x %d samples, oracle/ddc_oracle.c window form "
        "(gcc -O3 -march=native -fopenmp, 16 windows in the vector lanes, bit-exact vs the clock-level model, which runs "
        "%.0f MS/s on one thread); nproc reports %d CPUs, OMP_NUM_THREADS=%s"
        % (used, per_thread, samples, clock_one, os.cpu_count() or 0, os.environ.get("OMP_NUM_THREADS", "unset")))


def dry_run(args, rank, world):
    """The multi-rank path without a GPU: gloo ranks, the same StepGather the GPU path uses,
    a pattern that depends on (rank, step, position) in place of the kernels."""
    import torch
    import torch.distributed as dist
    from ua3reo_ddc import shard
    with _StdoutToStderr():  # gloo announces its connections on stdout
        dist.init_process_group("gloo", rank=rank, world_size=world)
    nch, nf = args.channels, max(1, args.samples // 1024)
    sg = shard.StepGather(dist, (2, nch, nf, 2), torch.int16, "cpu", timed=False)

    def pattern(r, k):
        base = torch.arange(2 * nch * nf * 2, dtype=torch.int64).reshape(2, nch, nf, 2)
        return ((base * 7 + 1000 * r + 31 * k) % 30011).to(torch.int16)

    total = args.warmup + args.steps
    dist.barrier()
    t0 = time.perf_counter()
    for k in range(total):
        sg.buffer(k).copy_(pattern(rank, k))
        sg.submit(k)
    sg.finish()
    dist.barrier()
    dt = time.perf_counter() - t0
    ok = all(torch.equal(sg.result(k)[r], pattern(r, k)) for k in range(max(0, total - 2), total) for r in range(world))
    if rank == 0:
        print(json.dumps({
            "metric": METRIC, "value": 0.0, "unit": "MS/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / max(1, total) * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "none", "data": "dry run on CPUs (gloo): a pattern stands in for the kernels",
            "config": {"workload": "dry run: %d channels x %d frames per rank" % (nch, nf),
                       "all_gather": {"bytes_per_rank": int(sg.outs[0].numel() * 2), "backend": "gloo",
                                      "gathered_equals_pattern": bool(ok)}}}), flush=True)
    dist.barrier()
    dist.destroy_process_group()
    return 0 if ok else 1


def measured_copy_gbs(torch, dev):
    """Device-to-device copy rate of this GPU (read + write bytes per second), the practical
    HBM figure next to the 8 TB/s of the data sheet: best of 5 copies of 1 GiB after a warm-up."""
    n = 1 << 30
    src = torch.empty(n, dtype=torch.uint8, device=dev)
    dst = torch.empty(n, dtype=torch.uint8, device=dev)
    src.zero_()
    dst.copy_(src)
    torch.cuda.synchronize()
    best = None
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        dst.copy_(src)
        e1.record()
        e1.synchronize()
        ms = e0.elapsed_time(e1)
        best = ms if best is None or ms < best else best
    del src, dst
    return 2.0 * n / (best * 1e-3) / 1e9


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse_args(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(args, argv)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    if args.dry_run_gloo:
        return dry_run(args, rank, world)

    import numpy as np
    import torch
    do_gather = world > 1 or args.force_gather
    if world > 1 or (args.force_gather and args.gather == "torch"):
        import torch.distributed as dist
        # stdout carries the one JSON line: RCCL's own banner / debug text goes to stderr
        os.environ.setdefault("NCCL_DEBUG_FILE", "/dev/stderr")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        torch.cuda.set_device(local_rank)
        with _StdoutToStderr():  # the communicator prints its version banner when it is built
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
            dist.barrier()
            torch.cuda.synchronize()
    else:
        dist = None
        torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    if args.workload == "tx":
        res = bench_tx(args, dev, local_rank, rank, world, dist)
    else:
        res = bench_rx(args, dev, local_rank, rank, world, dist, do_gather)
        if rank == 0 and world == 1 and args.workload == "ddc" and not args.no_secondary and not args.shared_stream:
            # BASELINE configs 3 and 4 at their shapes, a few steps each, timed by this same run
            sec = {}
            a3 = argparse.Namespace(**vars(args))
            a3.workload, a3.channels, a3.samples, a3.steps, a3.warmup = "ddc+fft", 256, 1 << 22, 5, 2
            a3.no_cpu_baseline = a3.no_copy_rate = True
            r3 = bench_rx(a3, dev, local_rank, rank, world, None, False)
            a4 = argparse.Namespace(**vars(a3))
            a4.workload = "tx"
            r4 = bench_tx(a4, dev, local_rank, rank, world, None)
            for name, r in (("config3", r3), ("config4", r4)):
                sec[name] = {"workload": r["config"]["workload"], "metric": r["metric"], "value": r["value"], "unit": r["unit"],
                             "steps": r["steps"], "ms_per_step": r["ms_per_step"], "frac": r["roofline"]["frac"],
                             "frac_of_step": r["roofline"]["frac_of_step"], "kernel": r["roofline"]["kernel"],
                             "parity_check": r["config"]["parity_check"],
                             "first_window_ms_per_step": r["config"]["clock_settle"].get("first_window_ms_per_step"),
                             "settle_calls": r["config"]["clock_settle"].get("calls")}
            res["secondary"] = sec
    if rank == 0:
        print(json.dumps(res), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    return 0


def settle_clock(step, sync, torch, dist, dev, enabled, chunk=20, max_chunks=40, tol=0.005):
    """Untimed calls of the same step until the time of a chunk of twenty stops changing
    (two chunks in a row within 0.5 % of the one before).

    After a load step the chip drops its clock within 2-3 ms and then raises it again over about
    30 ms of continuous load (profiles/r03_launch_series.txt: the ingest launch goes 628 -> 754 ->
    550 us and stays there for seconds); the driver's five warm-up steps end in the middle of that.
    The timed region that follows measures the rate the chip sustains, and the line reports this
    phase and the first window next to it.  Every rank runs the same number of chunks (they agree
    with one all-reduce per chunk), so steps that gather stay matched."""
    if not enabled:
        return {"calls": 0}
    prev, n, t_begin, first, calm = None, 0, time.perf_counter(), None, 0
    for _ in range(max_chunks):
        sync()
        t0 = time.perf_counter()
        for _ in range(chunk):
            step()
        sync()
        ms = (time.perf_counter() - t0) * 1e3 / chunk
        n += chunk
        first = ms if first is None else first
        calm = calm + 1 if prev is not None and abs(ms - prev) <= tol * prev else 0
        done = calm >= 2
        if dist is not None:
            flag = torch.tensor([0 if done else 1], dtype=torch.int32, device=dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)  # go on while any rank still changes
            done = not bool(flag.item())
        prev = ms
        if done:
            break
    return {"calls": n, "ms": round((time.perf_counter() - t_begin) * 1e3, 1),
            "first_chunk_ms_per_step": round(first, 4), "last_chunk_ms_per_step": round(prev, 4)}


def bench_rx(args, dev, local_rank, rank, world, dist, do_gather):
    """The RX chain (and, for ddc+fft, the panadapter rows) over resident ADC rows; returns the result dict on rank 0."""
    import numpy as np
    import torch
    import ua3reo_ddc
    from ua3reo_ddc import shard
    nch, S = args.channels, args.samples
    assert S % 1024 == 0
    nf = S // 1024
    rb = S // 2 * 3
    chan0 = rank * nch
    shared = args.shared_stream
    # synthetic rows: the library's generator (two tones + xorshift64* noise per channel), run on the GPU
    n_rows = 1 if shared else nch
    adc = torch.empty((n_rows, rb), dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    ua3reo_ddc.synth_adc_dev(chan0, n_rows, S, adc.data_ptr(), rb, FS_ADC)
    pitch = 0 if shared else rb
    with_fft = args.workload == "ddc+fft"
    if with_fft:
        assert nf % 512 == 0, "ddc+fft needs samples to be a multiple of 512*1024"
        fft_bins = torch.zeros((nch, nf // 512, 256), dtype=torch.float32, device=dev)
        fft_rows = torch.zeros((nch, nf // 512, 256), dtype=torch.int16, device=dev)
    phis = [ua3reo_ddc.phrase_from_frequency(1000000 + 48000 * (chan0 + c))[0] for c in range(nch)]
    if args.tune_all_hz is not None:  # experiment: at 0 Hz every lane fetches one NCO table entry (no LDS bank conflicts)
        phis = [ua3reo_ddc.phrase_from_frequency(args.tune_all_hz)[0]] * nch
    ctx = ua3reo_ddc.Ddc(nch, device=local_rank)
    ctx.tune_many(phis)
    if args.rx_plan == "split":
        ctx.debug_rx_plan(1)
    elif args.rx_plan == "valu":
        ctx.debug_rx_plan(2)
    fft_params = dict(mode="USB", fft_averaging=4)
    if with_fft:
        ctx.audio_configure(ua3reo_ddc.audio_params(**fft_params))
    ext = torch.cuda.ExternalStream(ctx.stream(), device=dev)
    # [buffer][SPEC|VOICE][channel][frame][I,Q]; two buffers so that the gather of step k
    # overlaps the compute of step k+1 (ua3reo_ddc/shard.py, StepGather)
    outs = torch.zeros((2, 2, nch, nf, 2), dtype=torch.int16, device=dev)
    gathered = None
    sg = None            # torch.distributed stand-in, only if the library's communicator cannot be built
    gather_path = None
    if do_gather:
        # the library's own collective (ddc_allgather_decimated_dev: one ncclAllGather of bytes on a second
        # stream, two slots); the communicator id travels over torch.distributed, which the ranks have anyway

        def all_ranks_ok(ok):
            if world == 1:
                return ok
            flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            return bool(flag.item())

        why = None
        try:
            if args.gather == "torch":
                raise RuntimeError("--gather torch")
            # every rank binds RCCL here, so that a rank without the library shows before anyone waits for it
            comm_id = ua3reo_ddc.comm_unique_id()
        except Exception as e:  # noqa: BLE001 (reported in the JSON line)
            why = "%s: %s" % (type(e).__name__, e)
        if all_ranks_ok(why is None):
            if world > 1:
                box = [comm_id if rank == 0 else None]
                dist.broadcast_object_list(box, src=0)
                comm_id = box[0]
            try:
                with _StdoutToStderr():
                    ctx.comm_init(world, rank, comm_id)
            except Exception as e:  # noqa: BLE001
                why = "%s: %s" % (type(e).__name__, e)
            if all_ranks_ok(why is None):
                gather_path = "library"
                gathered = torch.zeros((2, world, 2, nch, nf, 2), dtype=torch.int16, device=dev)
            elif why is None:
                ctx.comm_destroy()
                why = "ddc_comm_init failed on another rank"
        elif why is None:
            why = "RCCL could not be bound on another rank"
        if gather_path is None and dist is None:
            raise SystemExit("bench: the library's communicator could not be built: %s" % why)
        if gather_path is None:
            sys.stderr.write("bench: rank %d gathers over torch.distributed (%s)\n" % (rank, why))
            gather_path = "torch.distributed (the library's communicator could not be built: %s)" % why
            sg = shard.StepGather(dist, (2, nch, nf, 2), torch.int16, dev, compute_stream=ext, timed=True)
            outs, gathered = sg.outs, sg.gathered
    step_no = [0]
    torch.cuda.synchronize()

    def step():
        k = step_no[0]
        step_no[0] += 1
        if sg:
            o = sg.buffer(k)
        else:
            o = outs[k & 1]
            if do_gather:
                ctx.allgather_reuse(k & 1)  # the gather of step k - 2 has read this buffer
        ctx.rx_process_dev(adc.data_ptr(), pitch, S, o[0].data_ptr(), o[1].data_ptr())
        if with_fft:
            ctx.fft_process_dev(o[0].data_ptr(), nf, fft_bins.data_ptr(), fft_rows.data_ptr())
        if sg:
            sg.submit(k)
        elif do_gather:
            ctx.allgather_decimated_dev(k & 1, o.data_ptr(), nf, gathered[k & 1].data_ptr())

    # parity spot check before any timing: first call from power-on state vs the oracle
    check = "skipped"
    if not args.no_check:
        from oracle import oracle
        ns = min(S, 1 << 19) if with_fft else min(S, 1 << 17)  # 2^17 samples: the same call plan as the timed shape
        chk_ctx = ua3reo_ddc.Ddc(nch, device=local_rank)
        chk_ctx.tune_many(phis)
        chk_out = torch.zeros((2, nch, ns // 1024, 2), dtype=torch.int16, device=dev)
        torch.cuda.synchronize()
        chk_ctx.rx_process_dev(adc.data_ptr(), pitch, ns, chk_out[0].data_ptr(), chk_out[1].data_ptr())
        n_fft = ns // 1024 // 512 if with_fft else 0
        if n_fft:
            chk_ctx.audio_configure(ua3reo_ddc.audio_params(**fft_params))
            chk_bins = torch.zeros((nch, n_fft, 256), dtype=torch.float32, device=dev)
            chk_rows = torch.zeros((nch, n_fft, 256), dtype=torch.int16, device=dev)
            torch.cuda.synchronize()
            chk_ctx.fft_process_dev(chk_out[0].data_ptr(), n_fft * 512, chk_bins.data_ptr(), chk_rows.data_ptr())
        chk_ctx.sync()
        for c in sorted({0, nch // 2, nch - 1}):
            x = ua3reo_ddc.unpack_adc12(adc[0 if shared else c, : ns // 2 * 3].cpu().numpy(), ns)
            if c == 0 and not np.array_equal(x, ua3reo_ddc.synth_adc(chan0 + (0 if shared else c), ns, FS_ADC)):
                raise SystemExit("bench: device-generated ADC rows differ from the host generator")
            os_, ov_ = oracle.RxChain(phis[c]).run(x)
            if not (np.array_equal(chk_out[0, c].cpu().numpy(), os_) and
                    np.array_equal(chk_out[1, c].cpu().numpy(), ov_)):
                raise SystemExit("bench: GPU output differs from the oracle on channel %d" % c)
            if n_fft:
                a = oracle.Audio(oracle.audio_params(**fft_params))
                for f in range(n_fft):
                    wb, _, _ = a.fft(os_[512 * f:512 * (f + 1)])
                    if np.abs(chk_bins[c, f].cpu().numpy() - wb).max() > 1e-4 * max(float(wb.max()), 1e-3):
                        raise SystemExit("bench: panadapter bins differ from the oracle on channel %d" % c)
        chk_ctx.close()
        del chk_out
        check = "bit-exact vs oracle on 3 channels x %d samples%s" % (
            ns, "; %d panadapter rows each within 1e-4 of the frame maximum" % n_fft if n_fft else "")

    for _ in range(args.warmup):
        step()
    ctx.sync()
    if sg:
        sg.finish()
        if sg.times is not None:
            sg.times.clear()
    elif do_gather:
        ctx.allgather_wait(-1)
    torch.cuda.synchronize()

    def drain():
        ctx.sync()
        if sg:
            sg.finish()
        elif do_gather:
            ctx.allgather_wait(-1)
        torch.cuda.synchronize()

    # the window right behind the warm-up steps, timed the same way, for the record (the chip's clock is
    # still moving there); then untimed calls until it has settled; then the timed region proper
    first_window_ms = None
    if not args.no_settle:
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        drain()
        first_window_ms = (time.perf_counter() - t0) * 1e3 / args.steps
    settled = settle_clock(step, drain, torch, dist, dev, not args.no_settle)
    if sg and sg.times is not None:
        sg.times.clear()
    ctx.profile_enable(True)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    ctx.sync()
    gather_ms = []
    if sg:
        sg.finish()
        gather_ms = [sg.average_ms()] if sg.times else []
    elif do_gather:
        gather_ms = [ctx.allgather_wait(b) for b in range(min(2, args.steps))]
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    front_ms, tail_ms, ncalls = ctx.profile_collect()
    ctx.profile_enable(False)

    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if dist is not None:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())

    res = None
    if rank == 0:
        units = float(world) * nch * S * args.steps  # channel-samples ingested by the whole job
        value = units / dt / 1e6
        front_avg_ms = front_ms / max(1, ncalls)
        two_kernels = args.rx_plan != "split" and (S // 512) % 256 == 0
        kernel_name = ("ddc_rx_front_mx" if args.rx_plan == "auto" else "ddc_rx_front_cic") if two_kernels else "ddc_rx_front"
        achieved = BYTES_PER_CHANNEL_SAMPLE * nch * S / (front_avg_ms * 1e-3) / 1e9
        # the committed counter passes are of the default plan's ingest kernel
        traffic = None if shared or args.rx_plan != "auto" else load_profile("traffic.json", nch, S)
        valu = None if shared or args.rx_plan != "auto" else load_profile("valu.json", nch, S)
        step_ms = dt / args.steps * 1e3
        roof = {
            # the fraction is quoted against the HBM peak; what holds the kernel back is named below
            # once the vector-ALU counters of this kernel source are at hand
            "bound": "unknown (no vector-ALU counter pass on this kernel source; fraction quoted against the HBM peak)",
            "kernel": kernel_name,
            "achieved": round(achieved, 1),
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4),
            # the same algorithmic bytes over the whole step (every kernel of the call and the gaps between them)
            "frac_of_step": round(BYTES_PER_CHANNEL_SAMPLE * nch * S / (step_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            "traffic": float(traffic["traffic_bytes"]) if traffic else None,
            "traffic_source": ("%s @ commit %s" % (traffic["source"], traffic.get("commit", "?"))) if traffic else
                              "no PMC pass on this kernel source and shape in profiles/",
            "avg_launch_ms": round(front_avg_ms, 4),
            "rest_of_call_ms": round(tail_ms / max(1, ncalls), 4),
            "algorithmic_bytes_per_launch": BYTES_PER_CHANNEL_SAMPLE * nch * S,
        }
        if not args.no_copy_rate:
            copy_gbs = measured_copy_gbs(torch, dev)
            roof["measured_d2d_copy"] = {"gbs": round(copy_gbs, 1), "frac_of_it": round(achieved / copy_gbs, 4),
                                         "what": "torch copy of 1 GiB on this GPU after the timed region, read + write "
                                                 "bytes / best of 5; `frac` above stays against the 8000 GB/s peak"}
        if valu:
            # vector-ALU issue roof: instruction counts of the PMC pass priced with the probe's issue
            # cycles per instruction class, against the SIMD cycles of this run's launches
            waves = nch * S / 64.0
            need = (valu["slow_per_sample"] * valu["issue_cycles_slow"] +
                    valu["fast_per_sample"] * valu["issue_cycles_fast"]) * waves
            # SIMD cycles of one launch: counted in the PMC pass (GRBM_GUI_ACTIVE is summed over the 8 XCDs),
            # which does not depend on the clock the chip held; else this run's time at the probed clock
            have = N_SIMD * (valu["grbm_gui_active"] / 8.0 if valu.get("grbm_gui_active") else
                             front_avg_ms * 1e-3 * valu["clock_ghz"] * 1e9)
            # the clock this run held: the cycle count of a launch does not depend on it, its duration does
            clock_ghz = (valu["grbm_gui_active"] / 8.0 / (front_avg_ms * 1e-3) / 1e9) if valu.get("grbm_gui_active") \
                else valu["clock_ghz"]
            roof_s = need / (N_SIMD * clock_ghz * 1e9)  # launch time if the SIMDs never waited
            roof["valu"] = {
                "insts_per_sample": valu["insts_per_sample"],
                "slow_per_sample": valu["slow_per_sample"], "fast_per_sample": valu["fast_per_sample"],
                "issue_cycles_per_inst": {"slow": valu["issue_cycles_slow"], "fast": valu["issue_cycles_fast"]},
                "clock_ghz": round(clock_ghz, 3),
                "clock_ghz_of_counter_pass": valu["clock_ghz"],
                "issue_util": round(need / have, 4),
                "launch_ms_at_issue_roof": round(roof_s * 1e3, 4),
                "hbm_frac_at_issue_roof": round(BYTES_PER_CHANNEL_SAMPLE * nch * S / roof_s / 1e9 / HBM_PEAK_GBS, 4),
                "source": "%s @ commit %s" % (valu["source"], valu.get("commit", "?")),
            }
            if valu.get("issue_cycles_mixed"):
                # the two classes only reach their own rates in streams of one class; a stream with this
                # kernel's mix, every instruction independent, issues at one rate whatever the class
                # (probe rows front_mix_spread / front_mix_runs): the roof of the mixed stream
                need_mx = valu["insts_per_sample"] * valu["issue_cycles_mixed"] * waves
                roof_mx = need_mx / (N_SIMD * clock_ghz * 1e9)
                roof["valu"]["mixed_stream"] = {
                    "issue_cycles_per_inst": valu["issue_cycles_mixed"],
                    "issue_cycles_per_inst_fast_in_runs": valu.get("issue_cycles_mixed_runs"),
                    "issue_util": round(need_mx / have, 4),
                    "launch_ms_at_issue_roof": round(roof_mx * 1e3, 4),
                    "hbm_frac_at_issue_roof": round(BYTES_PER_CHANNEL_SAMPLE * nch * S / roof_mx / 1e9 / HBM_PEAK_GBS, 4),
                }
            roof["closer_roof"] = "valu-issue" if need / have > achieved / HBM_PEAK_GBS else "hbm"
            roof["bound"] = "hbm" if roof["closer_roof"] == "hbm" else \
                "valu-issue (vector-ALU instruction issue; fraction quoted against the HBM peak)"
        res = {
            "metric": METRIC,
            "value": round(value, 1),
            "unit": "MS/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "i32/i64 integer (12-bit ADC, 60-bit CIC)",
            "data": "synthetic: two tones + xorshift64* noise per channel (seed 0x5EED0000 + channel), packed 12-bit, "
                    "generated on the device by ddc_synth_adc_dev",
            "config": {
                "workload": "%s: %d channels x %d ADC samples per step, "
                            "full FPGA RX chain (NCO+mixer+rx_cic+rx_ciccomp+rx_hilb/delay) to SPEC+VOICE int16 I/Q%s"
                            % (("shared-stream fan-out (one ADC row, every channel its own tuning)" if shared else
                                "per-GPU shard of BASELINE config 5") if not with_fft else "BASELINE config 3 shape",
                               nch, S, " + 512-point FFT panadapter rows" if with_fft else ""),
                "channels_per_gpu": nch,
                "samples_per_channel": S,
                "fs_adc_hz": FS_ADC,
                "per_stream_msps": round(value / (world * nch), 1),
                "per_stream_times_real_time": round(value * 1e6 / (world * nch) / FS_ADC, 1),
                "parallelism": "channels sharded over %d GPU(s)%s" % (
                    world, ", RCCL all-gather of decimated I/Q each step, overlapped with the next step"
                    if do_gather else ""),
                "parity_check": check,
                "clock_settle": dict(settled, first_window_ms_per_step=None if first_window_ms is None else round(first_window_ms, 4),
                                     what="untimed calls of the same step between the warm-up steps and the timed region, until "
                                          "the time of twenty calls has changed by less than 0.5 % twice in a row: the chip lowers its clock right after "
                                          "load arrives and takes about 30 ms of load to raise it again; first_window = the "
                                          "--steps calls directly behind the warm-up, timed the same way"),
            },
            "roofline": roof,
        }
        if gather_ms:
            g_ms = sum(gather_ms) / len(gather_ms)
            k_last = step_no[0] - 1
            ok = bool(torch.equal(gathered[k_last & 1][rank], outs[k_last & 1]))
            res["config"]["all_gather"] = {"avg_ms": round(g_ms, 4),
                                           "of": "the timed gathers" if sg else "the last %d gathers" % len(gather_ms),
                                           "bytes_per_rank": int(outs[0].numel() * 2),
                                           "share_of_step": round(g_ms / (dt / args.steps * 1e3), 4),
                                           "own_shard_in_gathered": ok,
                                           "path": gather_path,
                                           "note": "ddc_allgather_decimated_dev: ncclAllGather on a second stream, "
                                                   "overlapped with the next step's compute" if not sg else
                                                   "all_gather_into_tensor on a second stream (ua3reo_ddc/shard.py), "
                                                   "overlapped with the next step's compute"}
        if not args.no_cpu_baseline and world == 1:
            want = int(os.environ.get("OMP_NUM_THREADS", "0") or 0) or (os.cpu_count() or 1)
            threads = max(1, min(want, os.cpu_count() or 1))
            v, one, used, sample = cpu_baseline(threads)
            res["cpu_baseline"] = {"value": round(v, 1), "unit": "MS/s", "cores": used,
                                   "kind": "port", "sample": sample, "one_thread": round(one, 1)}
    ctx.close()
    del adc, outs, gathered
    return res if rank == 0 else None


def bench_tx(args, dev, local_rank, rank, world, dist):
    """DUC: 48 kHz I/Q -> tx_ciccomp x2 -> tx_cic x512 -> mixer -> 14-bit DAC words.
    One step = args.samples DAC samples per channel; HBM traffic is the 2-byte DAC
    word written per sample plus 4 bytes read per 1024."""
    import numpy as np
    import torch
    import ua3reo_ddc
    nch, S = args.channels, args.samples
    assert S % 1024 == 0
    n_iq = S // 1024
    g = torch.Generator(device=dev).manual_seed(0x7E57 + rank)
    iq = torch.randint(-20000, 20000, (nch, n_iq, 2), device=dev, generator=g, dtype=torch.int32).to(torch.int16)
    dac = torch.zeros((nch, S), dtype=torch.int16, device=dev)
    chan0 = rank * nch
    phis = [ua3reo_ddc.phrase_from_frequency(1000000 + 48000 * (chan0 + c))[0] for c in range(nch)]
    ctx = ua3reo_ddc.Ddc(nch, device=local_rank)
    ctx.tune_many(phis)
    check = "skipped"
    if not args.no_check:
        from oracle import oracle
        k = min(n_iq, 8)
        # contiguous-row form: run k pairs of every channel from a compact copy
        iq_k = iq[:, :k].contiguous()
        dac_k = torch.zeros((nch, k * 1024), dtype=torch.int16, device=dev)
        chk = ua3reo_ddc.Ddc(nch, device=local_rank)
        chk.tune_many(phis)
        torch.cuda.synchronize()
        chk.tx_process_dev(iq_k.data_ptr(), k, dac_k.data_ptr())
        chk.sync()
        for c in sorted({0, nch // 2, nch - 1}):
            want = oracle.TxChain(phis[c]).run(iq_k[c].cpu().numpy())
            if not np.array_equal(dac_k[c].cpu().numpy().view(np.uint16), want):
                raise SystemExit("bench: GPU DAC stream differs from the oracle on channel %d" % c)
        chk.close()
        check = "bit-exact vs oracle on 3 channels x %d DAC samples" % (k * 1024)
    ev0 = torch.cuda.Event(enable_timing=True)
    ev1 = torch.cuda.Event(enable_timing=True)
    ext = torch.cuda.ExternalStream(ctx.stream(), device=dev)
    torch.cuda.synchronize()  # torch filled iq / dac on its own stream
    def step():
        ctx.tx_process_dev(iq.data_ptr(), n_iq, dac.data_ptr())

    def drain():
        ctx.sync()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    drain()
    first_window_ms = None
    if not args.no_settle:
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        drain()
        first_window_ms = (time.perf_counter() - t0) * 1e3 / args.steps
    settled = settle_clock(step, drain, torch, dist, dev, not args.no_settle)
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    ev0.record(ext)
    for _ in range(args.steps):
        ctx.tx_process_dev(iq.data_ptr(), n_iq, dac.data_ptr())
    ev1.record(ext)
    ctx.sync()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    call_ms = ev0.elapsed_time(ev1) / args.steps
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if dist is not None:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    res = None
    if rank == 0:
        bytes_per = 2.0 + 4.0 / 1024.0
        achieved = bytes_per * nch * S / (call_ms * 1e-3) / 1e9
        res = {
            "metric": "DAC MSamples/s produced per node, N-channel DUC",
            "value": round(float(world) * nch * S * args.steps / dt / 1e6, 1), "unit": "MS/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "i32/i64 integer (16-bit I/Q, 60-bit CIC, 14-bit DAC)",
            "data": "synthetic: uniform random 48 kHz I/Q, generated on device",
            "config": {"workload": "BASELINE config 4 shape: %d channels x %d I/Q pairs -> %d DAC words per channel "
                                   "per step (tx_ciccomp + tx_cic + tx_mixer + DAC_corrector)" % (nch, n_iq, S),
                       "channels_per_gpu": nch, "parity_check": check,
                       "clock_settle": dict(settled, first_window_ms_per_step=None if first_window_ms is None
                                            else round(first_window_ms, 4))},
            "roofline": {"bound": "valu-issue (vector-ALU instruction issue, profiles/r02_c_tx_sq_counters.txt; fraction "
                                  "quoted against the HBM peak)",
                         "kernel": "whole ddc_tx_process_dev call (ddc_tx_duc dominates)",
                         "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4),
                         "frac_of_step": round(bytes_per * nch * S / (dt / args.steps) / 1e9 / HBM_PEAK_GBS, 4),
                         "traffic": None, "avg_launch_ms": round(call_ms, 4)},
        }
    ctx.close()
    del iq, dac
    return res if rank == 0 else None


if __name__ == "__main__":
    sys.exit(main())
