#!/usr/bin/env python3
"""Turns the PMC summaries of one measurement round into the two small JSON files
bench.py replays next to its live timing (it cannot read counters from inside a run):

  profiles/traffic.json  HBM-side bytes per ddc_rx_front_mx launch (2 x FETCH_SIZE + WRITE_SIZE,
                         KiB -> bytes; gfx950 tallies 128-byte fills at 64 bytes)
  profiles/valu.json     vector instructions per ADC sample (SQ_INSTS_VALU), split into the two
                         issue classes the probe (tools/probe/valu_issue.hip) found, with the
                         probe's cycles per wave-instruction per SIMD and the clock of the pass

Both carry the hash of the kernel source they were measured on and the commit that holds it;
bench.py ignores them once csrc/ddc_rx_kernels.hip.h or csrc/ddc_rx_mx_kernel.hip.h has changed.  The tool refuses to write while
the kernel sources in the tree differ from HEAD: a counter pass is only worth recording against a
source that is in the history.

usage: tools/make_profile_json.py --sq gpurun_out/<tag>/pmc_sq.txt --traffic gpurun_out/<tag>/pmc_summary.txt
           --clock-ghz 2.03 [--channels 1024 --samples 1048576] --source "<what was run>"
"""
import argparse
import hashlib
import json
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRCS = [os.path.join(ROOT, "ua3reo-ddc-transceiver_amd", "csrc", f) for f in ("ddc_rx_kernels.hip.h", "ddc_rx_mx_kernel.hip.h")]
KERNEL = "ddc::ddc_rx_front_mx"
LOOP_TRIPS = 4.0  # the sample loop of ddc_rx_front_mx runs once per pass of 16 windows
# issue classes (profiles/r02_valu_issue_probe.txt): ~2.1 cycles per wave-instruction per SIMD
FAST = {"v_add_u32", "v_sub_u32", "v_subrev_u32", "v_lshrrev_b32", "v_ashrrev_i32", "v_and_b32", "v_or_b32",
        "v_xor_b32", "v_mov_b32", "v_fma_f32", "v_fmac_f32", "v_fmamk_f32", "v_fmaak_f32", "v_mul_f32", "v_add_f32",
        "v_mul_lo_u16", "v_add_u16"}
ISSUE_FAST, ISSUE_SLOW = 2.15, 4.07
# a stream with the kernel's class mix (22 : 10 in 32, all independent) at 3 waves per SIMD: the 2-cycle
# instructions spread among the others / in one run (probe rows front_mix_spread, front_mix_runs)
ISSUE_MIXED, ISSUE_MIXED_RUNS = 4.15, 3.90


def medians(path, kernel):
    out, cur = {}, None
    for line in open(path):
        if not line.startswith(" "):
            cur = line.strip()
            continue
        m = re.match(r"\s+(\w+)\s+n=\d+\s+max=\S+\s+median=(\S+)", line)
        if m and cur == kernel:
            out[m.group(1)] = float(m.group(2))
    return out


def static_mix():
    """Fractions (slow, fast) of the vector instructions of ddc_rx_front_mx (matrix instructions left
    out), blocks of the sample loop weighted by its trips, from the compiler's assembly of the source in
    the tree."""
    asm = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-o", "-",
                          os.path.join(ROOT, "ua3reo-ddc-transceiver_amd", "csrc", "ddc_lib.hip")],
                         stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True).stdout.decode()
    i = asm.index("\n_ZN3ddc15ddc_rx_front_mxE")
    body = asm[i:asm.index("s_endpgm", i)].split("\n")
    # the sample loop is the loop with the most instructions
    loops, cur = {}, None
    weight_of_line = []
    for ln in body:
        m = re.match(r"^(\.LBB\d+_\d+):\s*;?(.*)", ln)
        if m:
            note = m.group(2)
            h = re.search(r"Header=(BB\d+_\d+)", note)
            if "Loop Header" in note:
                cur = m.group(1)[2:]
            elif h:
                cur = h.group(1)
            else:
                cur = None
        weight_of_line.append(cur)
        if cur and re.match(r"^\s+v_", ln):
            loops[cur] = loops.get(cur, 0) + 1
    main = max(loops, key=loops.get)
    slow = fast = 0.0
    for ln, lp in zip(body, weight_of_line):
        m = re.match(r"^\s+(v_[a-z0-9_]+?)(_e32|_e64|_sdwa|_dpp)?\s", ln)
        if not m or m.group(1).startswith("v_mfma"):
            continue
        w = LOOP_TRIPS if lp == main else 1.0
        name, suffix = m.group(1), m.group(2)
        if name in FAST and suffix not in ("_sdwa", "_dpp"):
            fast += w
        else:
            slow += w
    return slow / (slow + fast), fast / (slow + fast), loops[main]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sq")
    ap.add_argument("--traffic")
    ap.add_argument("--clock-ghz", type=float)
    ap.add_argument("--channels", type=int, default=1024)
    ap.add_argument("--samples", type=int, default=1 << 20)
    ap.add_argument("--source", default="tools/pmc_sq.sh / tools/gpu_measure.sh")
    a = ap.parse_args()
    h = hashlib.sha256()
    for path in SRCS:
        h.update(open(path, "rb").read())
    sha = h.hexdigest()[:16]
    git = lambda *args: subprocess.run(["git", "-C", ROOT] + list(args), stdout=subprocess.PIPE, check=True).stdout.decode().strip()
    dirty = git("status", "--porcelain", "--", "ua3reo-ddc-transceiver_amd/csrc", "include", "bench.py")
    if dirty:
        raise SystemExit("make_profile_json: kernel sources / bench.py differ from HEAD, commit them first:\n" + dirty)
    commit = git("rev-parse", "--short", "HEAD")
    blob = " ".join(git("rev-parse", "HEAD:ua3reo-ddc-transceiver_amd/csrc/" + os.path.basename(path)) for path in SRCS)
    common = {"kernel": "ddc_rx_front_mx", "channels": a.channels, "samples": a.samples, "kernel_source_sha256_16": sha,
              "commit": commit, "kernel_source_git_blob": blob}
    if a.traffic:
        m = medians(a.traffic, KERNEL)
        t = dict(common, source=a.source + " (rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate runs)",
                 fetch_size_kb=m["FETCH_SIZE"], write_size_kb=m["WRITE_SIZE"],
                 correction="gfx950: FETCH_SIZE tallies 128-byte fills at 64 bytes, so reads are doubled; unit is KiB",
                 traffic_bytes=(2.0 * m["FETCH_SIZE"] + m["WRITE_SIZE"]) * 1024.0)
        json.dump(t, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)
        print("traffic.json:", t["traffic_bytes"])
    if a.sq:
        m = medians(a.sq, KERNEL)
        per_sample = m["SQ_INSTS_VALU"] * 64.0 / (a.channels * a.samples)
        fs, ff, loop_n = static_mix()
        v = dict(common, source=a.source + " (rocprofv3 --pmc SQ_INSTS_VALU ...; class split from the compiler's assembly, "
                 "sample loop of %d vector instructions per 128 samples weighted by its 4 trips; issue cycles from "
                 "tools/probe/valu_issue.hip, profiles/r02_valu_issue_probe.txt)" % loop_n,
                 sq_insts_valu=m["SQ_INSTS_VALU"], insts_per_sample=round(per_sample, 3),
                 slow_per_sample=round(per_sample * fs, 3), fast_per_sample=round(per_sample * ff, 3),
                 issue_cycles_slow=ISSUE_SLOW, issue_cycles_fast=ISSUE_FAST,
                 issue_cycles_mixed=ISSUE_MIXED, issue_cycles_mixed_runs=ISSUE_MIXED_RUNS, clock_ghz=a.clock_ghz,
                 sq_wave_cycles=m.get("SQ_WAVE_CYCLES"), sq_busy_cycles=m.get("SQ_BUSY_CYCLES"),
                 grbm_gui_active=m.get("GRBM_GUI_ACTIVE"))
        json.dump(v, open(os.path.join(ROOT, "profiles", "valu.json"), "w"), indent=1)
        print("valu.json:", v["insts_per_sample"], v["slow_per_sample"], v["fast_per_sample"])


if __name__ == "__main__":
    main()
