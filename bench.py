#!/usr/bin/env python3
"""Benchmark of the N-channel DDC ingest (BASELINE.json metric).

One step = one pass of the RX hot path (NCO mix, CIC /512, CIC compensator /2,
Hilbert / delay) over one batch of packed 12-bit ADC rows that already sit in
HBM.  Workload per GPU: the per-GPU shard of BASELINE config 5
(1024 tuned channels x 2^20 ADC samples each); with N GPUs the job is
1024*N channels (weak scaling) and every step ends with an RCCL all-gather of
the decimated SPEC+VOICE I/Q, as that config prescribes.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "ua3reo-ddc-transceiver_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

METRIC = "ADC MSamples/s ingested per node, N-channel DDC; % HBM roofline"
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak
BYTES_PER_CHANNEL_SAMPLE = 1.5 + 8.0 / 1024.0  # packed ADC read + SPEC/VOICE int16 I/Q written
FS_ADC = 49152000


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


def synth_rows(n_channels, samples, chan0, device, seed):
    """Packed 12-bit rows [n_channels][samples*3/2] on the device: per channel two
    tones (1400 and 300 LSB) at f_c = 1 MHz + 48 kHz*channel and +1 kHz, plus
    uniform noise in [-64, 63], clipped to +-2047."""
    rb = samples // 2 * 3
    out = torch.empty((n_channels, rb), dtype=torch.uint8, device=device)
    g = torch.Generator(device=device).manual_seed(seed)
    n = torch.arange(samples, device=device, dtype=torch.int64)
    step = 32
    for c0 in range(0, n_channels, step):
        c1 = min(n_channels, c0 + step)
        ch = torch.arange(chan0 + c0, chan0 + c1, device=device, dtype=torch.int64)[:, None]
        f1 = 1000000 + 48000 * ch
        ph1 = ((n[None, :] * f1) % FS_ADC).to(torch.float64) * (2.0 * np.pi / FS_ADC)
        ph2 = ((n[None, :] * (f1 + 1000)) % FS_ADC).to(torch.float64) * (2.0 * np.pi / FS_ADC)
        x = torch.round(1400.0 * torch.sin(ph1) + 300.0 * torch.sin(ph2)).to(torch.int32)
        del ph1, ph2
        x += torch.randint(-64, 64, x.shape, device=device, generator=g, dtype=torch.int32)
        x.clamp_(-2047, 2047)
        x &= 0xFFF
        s0, s1 = x[:, 0::2], x[:, 1::2]
        blk = out[c0:c1].view(c1 - c0, samples // 2, 3)
        blk[:, :, 0] = (s0 & 0xFF).to(torch.uint8)
        blk[:, :, 1] = ((s0 >> 8) | ((s1 & 0xF) << 4)).to(torch.uint8)
        blk[:, :, 2] = (s1 >> 4).to(torch.uint8)
        del x, s0, s1
    return out


def measured_traffic(nch, samples):
    """HBM-side bytes per ddc_rx_front launch from the last committed PMC passes
    (FETCH_SIZE doubled as gfx950 needs, plus WRITE_SIZE), if they were taken on
    this workload shape; rocprofv3 counters cannot be read from inside the run."""
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            t = json.load(f)
        if t["channels"] == nch and t["samples"] == samples:
            return float(t["traffic_bytes"])
    except (OSError, KeyError, ValueError):
        pass
    return None


def cpu_baseline(n_threads, samples=1 << 22, distinct=4, reps=24):
    """The CPU oracle (C restatement of the reference datapath, built here with
    gcc -O3 -march=native) on a bounded sample of the same workload: every thread
    runs the whole RX chain over `distinct` synthetic channels of `samples` samples,
    `reps` times.  The timed variant is the window form of mixer + rx_cic, whose lane
    loops the compiler vectorises (oracle/ddc_oracle.c, orc_fast_rx_front_run); it
    has to reproduce the clock-level model bit for bit before it is timed.  Also
    times one thread alone, and the clock-level model on one thread."""
    import tempfile
    import ua3reo_ddc
    from oracle import oracle
    L = oracle.build_native_variant(tempfile.mkdtemp(prefix="ddc_oracle_"))
    rows = [ua3reo_ddc.pack_adc12(ua3reo_ddc.synth_adc(c, samples)) for c in range(distinct)]
    phis = [ua3reo_ddc.phrase_from_frequency(1000000 + 48000 * c)[0] for c in range(distinct)]
    rb = len(rows[0])
    packed = np.concatenate(rows)
    # same answers as the clock-level model of the default build before any timing
    a = oracle.rx_chain_packed(packed, rb, distinct, 1 << 17, phis, L=L, fast=True)
    b = oracle.rx_chain_packed(packed, rb, distinct, 1 << 17, phis)
    if not (np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])):
        raise SystemExit("bench: the vectorised CPU chain disagrees with the clock-level oracle")

    def work(_):
        for _r in range(reps):
            oracle.rx_chain_packed(packed, rb, distinct, samples, phis, L=L, fast=True)

    t0 = time.perf_counter()
    oracle.rx_chain_packed(packed, rb, 1, samples, phis, L=L)
    clock_one = samples / (time.perf_counter() - t0) / 1e6
    best = 0.0
    for _r in range(3):
        t0 = time.perf_counter()
        oracle.rx_chain_packed(packed, rb, distinct, samples, phis, L=L, fast=True)
        best = max(best, distinct * samples / (time.perf_counter() - t0) / 1e6)
    t0 = time.perf_counter()
    with ThreadPoolExecutor(n_threads) as ex:
        list(ex.map(work, range(n_threads)))
    dt = time.perf_counter() - t0
    total = n_threads * reps * distinct * samples
    return total / dt / 1e6, best, "%d threads x %d channel-passes x %d samples, oracle/ddc_oracle.c window form " \
        "(gcc -O3 -march=native, lanes = 16 windows, bit-exact vs the clock-level model, which runs %.0f MS/s on one " \
        "thread); host reports %d CPUs" % (n_threads, reps * distinct, samples, clock_one, os.cpu_count() or 0)


def main():
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
    ap.add_argument("--shared-stream", action="store_true",
                    help="one ADC stream fanned out to every channel (row pitch 0) instead of one stream per channel; "
                         "reported apart from the headline: the ADC bytes then come from L2, not HBM")
    ap.add_argument("--force-gather", action="store_true",
                    help="run the per-step all-gather even with one rank (rehearses the N > 1 path on one GPU; "
                         "needs a torch.distributed launch)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 or args.force_gather:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
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

    import ua3reo_ddc
    if args.workload == "tx":
        return bench_tx(args, dev, local_rank, rank, world, dist)
    nch, S = args.channels, args.samples
    assert S % 1024 == 0
    nf = S // 1024
    rb = S // 2 * 3
    chan0 = rank * nch
    shared = args.shared_stream
    adc = synth_rows(1 if shared else nch, S, chan0, dev, 0x5EED0000 + rank)
    pitch = 0 if shared else rb
    # [buffer][SPEC|VOICE][channel][frame][I,Q]; two buffers so that the gather of
    # step k overlaps the compute of step k+1
    outs = torch.zeros((2, 2, nch, nf, 2), dtype=torch.int16, device=dev)
    out = outs[0]
    do_gather = world > 1 or args.force_gather
    gathered = torch.zeros((world * 2, nch, nf, 2), dtype=torch.int16, device=dev) if do_gather else None
    with_fft = args.workload == "ddc+fft"
    if with_fft:
        assert nf % 512 == 0, "ddc+fft needs samples to be a multiple of 512*1024"
        fft_bins = torch.zeros((nch, nf // 512, 256), dtype=torch.float32, device=dev)
        fft_rows = torch.zeros((nch, nf // 512, 256), dtype=torch.int16, device=dev)
    phis = [ua3reo_ddc.phrase_from_frequency(1000000 + 48000 * (chan0 + c))[0] for c in range(nch)]
    ctx = ua3reo_ddc.Ddc(nch, device=local_rank)
    ctx.tune_many(phis)
    if with_fft:
        ctx.audio_configure(ua3reo_ddc.audio_params(mode="USB", fft_averaging=4))
    ext = torch.cuda.ExternalStream(ctx.stream(), device=dev)
    comm = torch.cuda.Stream(device=dev) if do_gather else None
    computed = [torch.cuda.Event(), torch.cuda.Event()]
    gathered_ev = [torch.cuda.Event(), torch.cuda.Event()]
    step_no = [0]
    gather_times = None  # (start, end) events of every all-gather of the timed region
    torch.cuda.synchronize()

    def step():
        b = step_no[0] & 1
        step_no[0] += 1
        o = outs[b]
        if do_gather and step_no[0] > 2:
            ext.wait_event(gathered_ev[b])  # the gather that last read this buffer
        ctx.rx_process_dev(adc.data_ptr(), pitch, S, o[0].data_ptr(), o[1].data_ptr())
        if with_fft:
            ctx.fft_process_dev(o[0].data_ptr(), nf, fft_bins.data_ptr(), fft_rows.data_ptr())
        if do_gather:
            computed[b].record(ext)
            comm.wait_event(computed[b])
            with torch.cuda.stream(comm):
                if gather_times is not None:
                    t_a, t_b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    t_a.record(comm)
                dist.all_gather_into_tensor(gathered.view(torch.uint8), o.view(torch.uint8))
                if gather_times is not None:
                    t_b.record(comm)
                    gather_times.append((t_a, t_b))
                gathered_ev[b].record(comm)

    # parity spot check before any timing: first step from power-on state vs the oracle
    check = "skipped"
    if not args.no_check:
        from oracle import oracle
        ns = min(S, 1 << 16)
        chk_ctx = ua3reo_ddc.Ddc(nch, device=local_rank)
        chk_ctx.tune_many(phis)
        chk_out = torch.zeros((2, nch, ns // 1024, 2), dtype=torch.int16, device=dev)
        torch.cuda.synchronize()
        chk_ctx.rx_process_dev(adc.data_ptr(), pitch, ns, chk_out[0].data_ptr(), chk_out[1].data_ptr())
        chk_ctx.sync()
        for c in sorted({0, nch // 2, nch - 1}):
            x = ua3reo_ddc.unpack_adc12(adc[0 if shared else c, : ns // 2 * 3].cpu().numpy(), ns)
            os_, ov_ = oracle.RxChain(phis[c]).run(x)
            if not (np.array_equal(chk_out[0, c].cpu().numpy(), os_) and
                    np.array_equal(chk_out[1, c].cpu().numpy(), ov_)):
                raise SystemExit("bench: GPU output differs from the oracle on channel %d" % c)
        chk_ctx.close()
        del chk_out
        check = "bit-exact vs oracle on 3 channels x %d samples" % ns

    for _ in range(args.warmup):
        step()
    ctx.sync()
    torch.cuda.synchronize()
    ctx.profile_enable(True)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    if do_gather:
        gather_times = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    ctx.sync()
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

    if rank == 0:
        units = float(world) * nch * S * args.steps  # channel-samples ingested by the whole job
        value = units / dt / 1e6
        front_avg_ms = front_ms / max(1, ncalls)
        achieved = BYTES_PER_CHANNEL_SAMPLE * nch * S / (front_avg_ms * 1e-3) / 1e9
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
            "data": "synthetic: two tones + uniform noise per channel, packed 12-bit, generated on device",
            "config": {
                "workload": "%s: %d channels x %d ADC samples per step, "
                            "full FPGA RX chain (NCO+mixer+rx_cic+rx_ciccomp+rx_hilb/delay) to SPEC+VOICE int16 I/Q%s"
                            % (("shared-stream fan-out (one ADC row, every channel its own tuning)" if shared else
                                "per-GPU shard of BASELINE config 5") if not with_fft else "BASELINE config 3 shape",
                               nch, S, " + 512-point FFT panadapter rows" if with_fft else ""),
                "channels_per_gpu": nch,
                "samples_per_channel": S,
                "fs_adc_hz": FS_ADC,
                "parallelism": "channels sharded over %d GPU(s)%s" % (
                    world, ", RCCL all-gather of decimated I/Q each step, overlapped with the next step"
                    if world > 1 else ""),
                "parity_check": check,
            },
            "roofline": {
                "bound": "hbm",
                "kernel": "ddc_rx_front",
                "achieved": round(achieved, 1),
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4),
                "traffic": None if shared else measured_traffic(nch, S),
                "avg_launch_ms": round(front_avg_ms, 4),
                "rest_of_call_ms": round(tail_ms / max(1, ncalls), 4),
                "algorithmic_bytes_per_launch": BYTES_PER_CHANNEL_SAMPLE * nch * S,
            },
        }
        if gather_times:
            g_ms = sum(a.elapsed_time(b) for a, b in gather_times) / len(gather_times)
            res["config"]["all_gather"] = {"avg_ms": round(g_ms, 4), "bytes_per_rank": int(outs[0].numel() * 2),
                                           "share_of_step": round(g_ms / (dt / args.steps * 1e3), 4),
                                           "note": "on a second stream, overlapped with the next step's compute"}
        if not args.no_cpu_baseline and world == 1:
            threads = min(16, os.cpu_count() or 1)
            v, one, sample = cpu_baseline(threads)
            res["cpu_baseline"] = {"value": round(v, 1), "unit": "MS/s", "cores": threads,
                                   "kind": "port", "sample": sample, "one_thread": round(one, 1)}
        print(json.dumps(res), flush=True)
    ctx.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def bench_tx(args, dev, local_rank, rank, world, dist):
    """DUC: 48 kHz I/Q -> tx_ciccomp x2 -> tx_cic x512 -> mixer -> 14-bit DAC words.
    One step = args.samples DAC samples per channel; HBM traffic is the 2-byte DAC
    word written per sample plus 4 bytes read per 1024."""
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
    for _ in range(args.warmup):
        ctx.tx_process_dev(iq.data_ptr(), n_iq, dac.data_ptr())
    ctx.sync()
    torch.cuda.synchronize()
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
                       "channels_per_gpu": nch, "parity_check": check},
            "roofline": {"bound": "hbm", "kernel": "whole ddc_tx_process_dev call (ddc_tx_duc dominates)",
                         "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None,
                         "avg_launch_ms": round(call_ms, 4)},
        }
        print(json.dumps(res), flush=True)
    ctx.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
