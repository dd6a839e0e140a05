"""From the FETCH_SIZE / WRITE_SIZE summaries of tools/profile_round.sh (pmc_b64_{fetch,write}.txt for the headline step,
pmc_c5_b64_{fetch,write}.txt for config 5) build the two files bench.py scales into `roofline.traffic`:
    pmc_conv_block2_fwd_f16.json   HBM bytes per launch of the block-2 forward convolution at bs 64
    pmc_mrstft.json                HBM bytes per mx_mrstft_loss call (all its kernels) at bs 64
Counter units and the gfx950 correction as MI355X_MICROARCH.md's HBM / rocprofv3 section prescribes: FETCH_SIZE and WRITE_SIZE
in KB of 1000 B; FETCH_SIZE x 2 for kernels whose reads are 16 B per lane (incl. LDS-DMA) -- the convolution.  The MR-STFT
kernels read 4 B per lane, a width the guide leaves uncalibrated, so the factor is calibrated on a known byte count in this
very access pattern: mr_fold_all_kernel reads 271 MB of run sums + ~69 MB of run tails at bs 64 (three resolutions, two
components) and FETCH_SIZE reports 179.5 MB -> x 1.9: the same 1/2 (128-byte requests tallied at 64 B); x 2 is applied.
    python tools/pmc_traffic_json.py profiles/r04
"""
import json
import os
import re
import sys

d = sys.argv[1]


def per_launch(path, pattern, counter):
    """sum over the kernels matching `pattern` of (counter per launch x launches per step)."""
    out = {}
    for line in open(path):
        m = re.match(r"(.{60}) x\s*(\d+)\s+([\d.]+) ms/launch\s+(.*)", line)
        if not m or not re.search(pattern, m.group(1)):
            continue
        vals = dict(kv.split("=") for kv in m.group(4).split())
        out[m.group(1).strip()] = (int(m.group(2)), float(vals[counter]))
    return out


conv_f = per_launch(os.path.join(d, "pmc_b64_fetch.txt"), r"conv_f16x3_dma16_kernel<1, (true|false)>", "FETCH_SIZE")
conv_w = per_launch(os.path.join(d, "pmc_b64_write.txt"), r"conv_f16x3_dma16_kernel<1, (true|false)>", "WRITE_SIZE")
if conv_f and conv_w:
    f = list(conv_f.values())[0][1]
    w = list(conv_w.values())[0][1]
    alg = 64 * (2 * 128 * 4 * 352 * 16 * 2 + 64 * 64 * 352 * 4 + 64 * 64 * 352 + 64 * 64 * 2 * 4)
    json.dump({"kernel": "conv_f16x3_dma16_kernel<1, true>", "batch_measured": 64, "fetch_size_kb": f, "write_size_kb": w,
               "fetch_correction": 2.0, "hbm_bytes_per_launch_b64": (2.0 * f + w) * 1000.0, "algorithmic_bytes_b64": alg,
               "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes (pmc_b64_fetch.txt, pmc_b64_write.txt), KB = 1000 B, "
                       "gfx950 FETCH_SIZE x2 correction for 16 B/lane loads incl. LDS-DMA; algorithmic = operand pair read once + pooled "
                       "fp32 output, argmax bytes and the per-row LayerNorm partial sums written once; scale linearly with batch"},
              open(os.path.join(d, "pmc_conv_block2_fwd_f16.json"), "w"), indent=1)
    print("conv:", (2.0 * f + w) * 1000.0 / alg, "x algorithmic")
p5f, p5w = os.path.join(d, "pmc_c5_b64_fetch.txt"), os.path.join(d, "pmc_c5_b64_write.txt")
if os.path.exists(p5f) and os.path.exists(p5w):
    mf = per_launch(p5f, r"mr_", "FETCH_SIZE")
    mw = per_launch(p5w, r"mr_", "WRITE_SIZE")
    steps = 3                                            # bench.py --steps 2 --warmup 1: three calls of mx_mrstft_loss
    fetch = sum(n * v for n, v in mf.values()) / steps
    write = sum(n * v for n, v in mw.values()) / steps
    alg = 64 * 176400 * 12
    json.dump({"kernel": "mx_mrstft_loss (mr_onepass_kernel x 3 + mr_fold_all_kernel + finish)", "batch_measured": 64,
               "fetch_size_kb": fetch, "write_size_kb": write, "fetch_correction": 2.0,
               "hbm_bytes_per_launch_b64": (2.0 * fetch + write) * 1000.0, "algorithmic_bytes_b64": alg,
               "per_kernel_kb_per_launch": {k: {"launches": n, "fetch_kb": v, "write_kb": mw.get(k, (0, 0.0))[1]} for k, (n, v) in mf.items()},
               "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over bench.py --config 5 --batch 64 (pmc_c5_b64_*.txt), "
                       "KB = 1000 B, summed over the kernels of one mx_mrstft_loss call; FETCH_SIZE x2 (calibrated on mr_fold_all_kernel's known "
                       "340 MB of reads against 179.5 MB counted: the gfx950 half-count holds for these 4 B/lane streaming reads too); "
                       "algorithmic = 12 B per sample (x, y in, gradient out); the rest is the two gradient components' run sums, written "
                       "and read once"},
              open(os.path.join(d, "pmc_mrstft.json"), "w"), indent=1)
    print("mrstft:", (2.0 * fetch + write) * 1000.0 / alg, "x algorithmic")

# vector-pipe counters of the MR-STFT kernels: pmc_c5_b64_valu.txt -> pmc_mrstft_valu.json
p5v = os.path.join(d, "pmc_c5_b64_valu.txt")
if os.path.exists(p5v):
    per = {}
    tot = {"SQ_INSTS_VALU": 0.0, "SQ_ACTIVE_INST_VALU": 0.0, "SQ_BUSY_CU_CYCLES": 0.0, "SQ_WAVE_CYCLES": 0.0}
    for line in open(p5v):
        m = re.match(r"(.{60}) x\s*(\d+)\s+([\d.]+) ms/launch\s+(.*)", line)
        if not m or "mr_onepass" not in m.group(1):
            continue
        vals = {k: float(v) for k, v in (kv.split("=") for kv in m.group(4).split())}
        per[m.group(1).strip()] = dict(vals, ms_per_launch=float(m.group(3)),
                                       valu_busy=round(vals["SQ_ACTIVE_INST_VALU"] / vals["SQ_BUSY_CU_CYCLES"], 4),
                                       valu_share_of_wave_cycles=round(vals["SQ_ACTIVE_INST_VALU"] / vals["SQ_WAVE_CYCLES"], 4))
        for k in tot:
            tot[k] += vals[k]
    if per:
        json.dump({"kernels": per, "batch_measured": 64,
                   "valu_busy": round(tot["SQ_ACTIVE_INST_VALU"] / tot["SQ_BUSY_CU_CYCLES"], 4),
                   "valu_insts_per_call_b64": tot["SQ_INSTS_VALU"],
                   "note": "rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES over bench.py --config 5 --batch 64; "
                           "valu_busy = SQ_ACTIVE_INST_VALU (quad-cycles over all waves) / SQ_BUSY_CU_CYCLES (cycles over all CUs) = the "
                           "fraction of the four SIMDs' vector issue time in use (1.0 on the bare phaser cascade probe); "
                           "SQ_INSTS_VALU is per wave-instruction (a packed fp32 FMA = 4 flop x 64 lanes)"},
                  open(os.path.join(d, "pmc_mrstft_valu.json"), "w"), indent=1)
        print("mrstft valu busy:", tot["SQ_ACTIVE_INST_VALU"] / tot["SQ_BUSY_CU_CYCLES"])
