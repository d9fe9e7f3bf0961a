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

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak
BYTES_PER_CHANNEL_SAMPLE = 1.5 + 8.0 / 1024.0  # packed ADC read + SPEC/VOICE int16 I/Q written
FS_ADC = 49152000


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


def cpu_baseline(n_threads, samples=1 << 22, distinct=4, reps=6):
    """The CPU oracle (scalar C restatement of the reference datapath) on a
    bounded sample of the same workload: every thread runs the whole RX chain
    over `distinct` synthetic channels of `samples` samples, `reps` times."""
    import ua3reo_ddc
    from oracle import oracle
    oracle.lib()
    rows = [ua3reo_ddc.pack_adc12(ua3reo_ddc.synth_adc(c, samples)) for c in range(distinct)]
    phis = [ua3reo_ddc.phrase_from_frequency(1000000 + 48000 * c)[0] for c in range(distinct)]
    rb = len(rows[0])
    packed = np.concatenate(rows)

    def work(_):
        for _r in range(reps):
            oracle.rx_chain_packed(packed, rb, distinct, samples, phis)

    oracle.rx_chain_packed(packed, rb, 1, 1 << 16, phis)  # warm-up (tables, page faults)
    t0 = time.perf_counter()
    with ThreadPoolExecutor(n_threads) as ex:
        list(ex.map(work, range(n_threads)))
    dt = time.perf_counter() - t0
    total = n_threads * reps * distinct * samples
    return total / dt / 1e6, "%d threads x %d channel-passes x %d samples, oracle/ddc_oracle.c (gcc -O2, scalar)" % (
        n_threads, reps * distinct, samples)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--channels", type=int, default=1024, help="channels per GPU")
    ap.add_argument("--samples", type=int, default=1 << 20, help="ADC samples per channel per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-check", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        dist = None
        torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    import ua3reo_ddc
    nch, S = args.channels, args.samples
    assert S % 1024 == 0
    nf = S // 1024
    rb = S // 2 * 3
    chan0 = rank * nch
    adc = synth_rows(nch, S, chan0, dev, 0x5EED0000 + rank)
    out = torch.zeros((2, nch, nf, 2), dtype=torch.int16, device=dev)  # [SPEC|VOICE]
    gathered = torch.zeros((world,) + tuple(out.shape), dtype=torch.int16, device=dev) if world > 1 else None
    phis = [ua3reo_ddc.phrase_from_frequency(1000000 + 48000 * (chan0 + c))[0] for c in range(nch)]
    ctx = ua3reo_ddc.Ddc(nch, device=local_rank)
    ctx.tune_many(phis)
    ext = torch.cuda.ExternalStream(ctx.stream(), device=dev)
    torch.cuda.synchronize()

    def step():
        ctx.rx_process_dev(adc.data_ptr(), rb, S, out[0].data_ptr(), out[1].data_ptr())
        if world > 1:
            with torch.cuda.stream(ext):
                dist.all_gather_into_tensor(gathered, out)

    # parity spot check before any timing: first step from power-on state vs the oracle
    check = "skipped"
    if not args.no_check:
        from oracle import oracle
        ns = min(S, 1 << 16)
        chk_ctx = ua3reo_ddc.Ddc(nch, device=local_rank)
        chk_ctx.tune_many(phis)
        chk_out = torch.zeros((2, nch, ns // 1024, 2), dtype=torch.int16, device=dev)
        chk_ctx.rx_process_dev(adc.data_ptr(), rb, ns, chk_out[0].data_ptr(), chk_out[1].data_ptr())
        chk_ctx.sync()
        for c in sorted({0, nch // 2, nch - 1}):
            x = ua3reo_ddc.unpack_adc12(adc[c, : ns // 2 * 3].cpu().numpy(), ns)
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
            "metric": "ADC MSamples/s ingested per node, N-channel DDC",
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
                "workload": "per-GPU shard of BASELINE config 5: %d channels x %d ADC samples per step, "
                            "full FPGA RX chain (NCO+mixer+rx_cic+rx_ciccomp+rx_hilb/delay) to SPEC+VOICE int16 I/Q"
                            % (nch, S),
                "channels_per_gpu": nch,
                "samples_per_channel": S,
                "fs_adc_hz": FS_ADC,
                "parallelism": "channels sharded over %d GPU(s)%s" % (
                    world, ", RCCL all-gather of decimated I/Q each step" if world > 1 else ""),
                "parity_check": check,
            },
            "roofline": {
                "bound": "hbm",
                "kernel": "ddc_rx_front",
                "achieved": round(achieved, 1),
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4),
                "traffic": None,
                "avg_launch_ms": round(front_avg_ms, 4),
                "rest_of_call_ms": round(tail_ms / max(1, ncalls), 4),
                "algorithmic_bytes_per_launch": BYTES_PER_CHANNEL_SAMPLE * nch * S,
            },
        }
        if not args.no_cpu_baseline:
            threads = min(16, os.cpu_count() or 1)
            v, sample = cpu_baseline(threads)
            res["cpu_baseline"] = {"value": round(v, 1), "unit": "MS/s", "cores": threads,
                                   "kind": "port", "sample": sample}
        print(json.dumps(res), flush=True)
    ctx.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
