#!/usr/bin/env python3
"""Benchmarks of the hot path on MI355X, one process per GPU.  `--config` picks the BASELINE.json configuration:

  3 (default, the headline)  train_lfo_interwoven_all: 2D-CNN LFO extractor TRAIN STEP, bs = 256 x 2 s per GPU,
                             flanger / chorus / phaser interleaved
  2                          train_lfo_phaser: the same step on an all-phaser batch, bs = 64 x 2 s
  4                          train_em_dry_wet: frozen LFO-CNN + LSTM-64 effect model, truncated BPTT (83 optimizer steps
                             per batch), bs = 128 x 2 s
  5                          large-batch stress: flanger render + multi-resolution STFT loss forward / backward,
                             bs = 256 x 4 s per GPU (2048 x 4 s over 8 GPUs)

    python bench.py --gpus N --steps K --warmup W [--config 3]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Outside torchrun the first form is a LAUNCHER: it never touches the GPU itself, starts the N rank processes (fresh
children; through torch.distributed.run when N > 1) and relays rank 0's result.  At N = 1 on the headline config it
then measures configs 2 / 4 / 5 the same way (one fresh process each) and appends them as `other_configs`, so a single
driver run carries every BASELINE configuration.  `--worker` runs one measurement in the calling process (use that
form under rocprofv3).

OUTPUT (round 6).  The LAST stdout line is ONE short JSON object (< 6 KB: `compact()` below) -- metric, value, unit,
n_gpus, steps, warmup, ms_per_step, dtype, config, `roofline` (numbers + one short note), `cpu_baseline`, step statistics,
floor fractions of the sample-recurrent kernels, and a five-number summary per other config.  Everything else (per-kernel
tables, notes, floors, the other configs in full) goes to `bench_detail.json` next to this file (`--detail-out PATH`) and,
as one `BENCH_DETAIL {...}` line, to stderr.  N > 1: the line also carries `allreduce_ms`, `step_ms_per_rank`,
`rank_median_spread_ms`.

One step of config 2 / 3 = on-device batch synthesis (LFO synth, flanger/chorus, phaser; inputs are generated in
HBM) -> log-mel -> 6-block 2D-CNN forward -> L1 + 5*FDL1 + 10*SDL1 loss -> backward -> gradient all-reduce (RCCL)
-> AdamW.  Nothing is skipped or cached inside the timed region.  Rank 0 prints ONE JSON line.  Metric of every
config: 44.1 kHz audio-seconds per second of wall clock, whole job.

Extra objects in the JSON line:
  roofline      the governing kernel of the config.  Config 2 / 3: conv_f16x3_dma_kernel<1,0> = block-2 forward (5x13
                conv + bias + max-pool on split-fp16 operands, fp32-equivalent); ALGORITHMIC flops of that launch / its
                HIP-event duration measured live on the launch stream, vs the 2516.6 TFLOP/s dense fp16 MFMA peak (the
                pipes execute 3 MFMAs per algorithmic MAC group: achieved_executed / frac_executed).  Config 4: the LSTM
                forward; config 5: the MR-STFT kernels (HBM).
  kernels       the other measured launches.  The sample-recurrent kernels (flanger, phaser, LSTM forward / backward)
                carry, next to their algorithmic HBM rate (SURVEY.md 8d bytes / HIP-event duration / 8 TB/s), a MEASURED
                SERIAL FLOOR: the same launch through its `*_probe` twin entry point -- identical LDS traffic and dependent chain,
                no global-memory traffic inside the loop -- and `frac_of_serial_floor` = floor / real duration.
                north_star's "fx.py recurrent kernel >= 60 % of its measured roofline" is `fx_kernel_frac_of_serial_floor`
                (the flanger / chorus launch): for a kernel whose HBM time is 1 % of its dependency chain the chain,
                not HBM, is the roofline that governs, and both fractions are printed.
  step_ms       min / median / max over the timed steps (HIP events on the main stream)
  exact_fp32_path  (config 2 / 3, N = 1) the same step with the exact-fp32 MFMA convolutions, 3 steps
  cpu_baseline  the CPU oracle (oracle/: torch fp32 + C effects, "port") timed on this host's cores on a bounded
                sample (rank 0, N = 1 only); `reference_shaped` inside it = SURVEY 8d's form (B = 16, the flanger as the
                reference's python loop per sample, timed on a short stretch and extrapolated).
  other_configs (N = 1, config 3) value / ms / roofline / floors of configs 2, 4, 5, each measured in its own process.
"""
import argparse
import json
import os
import statistics
import sys
import time

torch = None        # imported by the worker only: the launcher process must never initialise the GPU
# multi-process GPU work on this pool needs dmabuf IPC (RCCL / sharing device tensors across processes fail with
# `hipIpcGetMemHandle: invalid argument` otherwise); the image exports it already -- kept here for any environment the ranks are built in
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SR, N_SAMPLES = 44100, 88200
LOSS = {"l1": 1.0, "fdl1": 5.0, "sdl1": 10.0, "mse": 0.0}
CNN_CFG = dict(in_ch=2, n_samples=N_SAMPLES, sr=SR, n_fft=1024, hop_len=256, n_mels=256, kernel_size=(5, 13),
               out_channels=[64] * 6, temp_dilations=[1, 1, 2, 4, 8, 16], pool_size=(2, 1), latent_dim=1,
               freq_mask_amount=0.25, time_mask_amount=0.25, use_ln=True)       # configs/models/spectral_2dcnn.yml
FP32_MFMA_PEAK_TFLOPS = 157.3                                                   # MI355X_MICROARCH.md
F16_MFMA_PEAK_TFLOPS = 2516.6                                                   # dense fp16/bf16 MFMA = 16 x the fp32 rate
HBM_PEAK_GBPS = 8000.0                                                          # MI355X_MICROARCH.md
W_FRAMES = N_SAMPLES // 256 + 1
# useful flops of one conv launch: 2 * Cout * Cin * 65 taps * H * W per clip
BLOCK_H = [256, 128, 64, 32, 16, 8]
BLOCK_CIN = [2, 64, 64, 64, 64, 64]
CONFIGS = {
    2: dict(name="train_lfo_phaser", kinds=("phaser",), batch=64, seconds=2.0),
    3: dict(name="train_lfo_interwoven_all", kinds=("flanger", "chorus", "phaser"), batch=256, seconds=2.0),
    4: dict(name="train_em_dry_wet", kinds=("phaser",), batch=128, seconds=2.0),
    5: dict(name="stress_flanger_mrstft", kinds=("flanger",), batch=256, seconds=4.0),
}
METRIC = {2: "44.1 kHz audio-seconds/sec (train step), phaser",
          3: "44.1 kHz audio-seconds/sec (train step), interwoven ph/fl/ch",
          4: "44.1 kHz audio-seconds/sec (TBPTT effect-model train batch), phaser pairs",
          5: "44.1 kHz audio-seconds/sec (flanger render + MR-STFT loss fwd/bwd), 4 s clips"}


def measured_traffic(batch: int, kind: str = "f32"):
    """HBM bytes per launch of the roofline kernel from the latest committed PMC pass (profiles/rNN/
    pmc_conv_block2_fwd[_f16].json, pmc_mrstft.json; FETCH_SIZE / WRITE_SIZE collected in separate rocprofv3 --pmc runs
    and corrected as MI355X_MICROARCH.md prescribes), scaled linearly to this batch; None if absent."""
    import glob
    name = {"f16": "pmc_conv_block2_fwd_f16.json", "mrstft": "pmc_mrstft.json"}.get(kind, "pmc_conv_block2_fwd.json")
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", name)))
    if not files:
        return None
    d = json.load(open(files[-1]))
    return d["hbm_bytes_per_launch_b64"] * batch / d["batch_measured"]


def measured_valu():
    """Vector-pipe counters of the MR-STFT kernels from the latest committed PMC pass (profiles/rNN/pmc_mrstft_valu.json:
    rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES over bench.py --config 5 --batch 64); None if absent."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "pmc_mrstft_valu.json")))
    return json.load(open(files[-1])) if files else None


MR_RESOLUTIONS = ((1024, 120), (2048, 240), (512, 50))       # auraloss defaults (n_fft, hop), oracle/losses.py
MR_FLOP_PER_BIN = 40                                          # Hermitian separation, |X|, |Y|, logs, loss sums, G1, G2, completion


def mrstft_flops_per_clip(n_samples: int):
    """Nominal fp32 flop count of the value + gradient of one clip: per frame TWO complex transforms (forward of x + i y,
    and the frame's share of the paired inverse transforms of G1, G2) at the textbook 5 N log2 N each, MR_FLOP_PER_BIN per bin
    of the N / 2 + 1, 6 N for windowing both signals and the two overlap-adds.  (The radix-4 kernels execute ~15 % fewer
    butterfly flops than 5 N log2 N; the count is the algorithm's, not the instruction stream's.)"""
    import math
    total, per_frame = 0.0, {}
    for n_fft, hop in MR_RESOLUTIONS:
        frames = n_samples // hop + 1
        f = 2 * 5 * n_fft * math.log2(n_fft) + (n_fft // 2 + 1) * MR_FLOP_PER_BIN + 6 * n_fft
        per_frame[str(n_fft)] = {"frames": frames, "flop_per_frame": int(f)}
        total += frames * f
    return total, per_frame


def conv_flops(block: int, batch: int) -> float:
    return 2.0 * 64 * BLOCK_CIN[block] * 65 * BLOCK_H[block] * W_FRAMES * batch


def hbm_block(kernel: str, bytes_alg: float, ms: float, floor_ms=None, note=None, bound="hbm"):
    """Roofline block of an HBM-nominal kernel: ALGORITHMIC bytes per launch / HIP-event duration.
    bound = "latency" (the sample-recurrent kernels K2 / K3 / K10): achieved / peak / frac stay the NOMINAL HBM figures
    (SURVEY 8d asks for them), the governing bound is the dependency chain -- `frac_of_independent_floor` (a floor built from a
    stand-alone microbenchmark, not from the kernel) and `frac_of_serial_floor` (the kernel's own probe twin) say how close
    the launch is to it."""
    gbps = bytes_alg / (ms * 1e-3) / 1e9
    out = {"bound": bound, "kernel": kernel, "algorithmic_bytes": int(bytes_alg), "avg_launch_ms": round(ms, 4),
           "achieved": round(gbps, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(gbps / HBM_PEAK_GBPS, 5),
           "traffic": None}
    if bound == "latency":
        out["nominal_bound"] = "hbm"
        out["bound_note"] = ("latency-bound: one workgroup per clip, the launch lasts as long as its chain of dependent steps; "
                             "achieved / peak / frac are the nominal HBM figures, the governing fractions are "
                             "frac_of_independent_floor / frac_of_serial_floor")
    if floor_ms is not None:
        out["serial_floor_ms"] = round(floor_ms, 4)
        out["frac_of_serial_floor"] = round(floor_ms / ms, 4)
    if note:
        out["note"] = note
    return out


def mean(v):
    return sum(v) / len(v)


def sync_replicas(opt):
    """DDP: rank 0's initial parameters on every rank (the seeds already agree; this makes it hold by construction)."""
    if torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1:
        torch.distributed.broadcast(opt.flat_param, src=0)


def build_lfo_job(device, rank, batch, kinds, overlap=True):
    from mod_extraction_amd import data_modules, lightning, models, optim
    import numpy as np
    torch.manual_seed(43)                 # the replicas start from the same weights on every rank ...
    np.random.seed(43)
    model = models.Spectral2DCNN(**CNN_CFG)
    module = lightning.LFOExtraction(model, sr=SR, use_dry=True, model_smooth_n_frames=0, should_stretch=False,
                                     loss_dict=LOSS).to(device)
    module.train()
    opt = optim.FlatAdamW(module.parameters(), lr=1e-4, betas=(0.8, 0.99))
    sync_replicas(opt)
    torch.manual_seed(43 + rank)          # ... and draw their own clips
    np.random.seed(43 + rank)
    batcher = data_modules.SyntheticFxBatcher(batch, N_SAMPLES, SR, kinds, device, audio_seed=43 + rank,
                                              overlap=overlap)
    return module, opt, batcher


def cpu_baseline_lfo(kinds, batch_cpu: int = 8, steps: int = 5, reference_shaped: bool = True):
    """The CPU oracle's version of the same step on the host cores, bounded sample (~10 s).
    Thread count: torch's CPU conv stops scaling at ~32 threads for this batch (measured on the 256-core
    GPU host: 16 thr 14.5, 32 thr 16.2, 64 thr 9.4, 256 thr 0.5 audio-s/s), so min(32, cores) is used
    and reported as `cores`."""
    import numpy as np
    from oracle import lightning as ol, models as om
    cores = min(32, os.cpu_count() or 1)
    torch.set_num_threads(cores)
    torch.manual_seed(43)
    np.random.seed(43)
    model = om.Spectral2DCNN(**CNN_CFG)
    model.train()
    opt = torch.optim.AdamW(model.parameters(), lr=1e-4, betas=(0.8, 0.99))
    from mod_extraction_amd.data_modules import SyntheticFxBatcher
    sampler = SyntheticFxBatcher(batch_cpu, N_SAMPLES, SR, kinds, torch.device("cpu"))   # host-side parameter draws only
    times = []
    for it in range(steps + 1):
        t0 = time.perf_counter()
        p = sampler.sample_params()
        src = (torch.rand(batch_cpu, N_SAMPLES + sampler.max_lead) * 2 - 1).mul_(sampler.peak).numpy()
        dry, wet, mod = ol.synth_batch(p, sampler.kinds, src, N_SAMPLES, SR, {"flanger": 1.0, "chorus": 30.0})
        ol.lfo_train_step(model, opt, dry, wet, mod, LOSS)
        times.append(time.perf_counter() - t0)
    t = sum(times[1:]) / steps
    out = {"value": batch_cpu * N_SAMPLES / SR / t, "unit": "audio-seconds/s", "cores": cores, "kind": "port",
           "sample": f"{steps} train steps (after 1 warm-up) of the CPU oracle on {batch_cpu} clips x 2 s, same "
                     f"{'/'.join(kinds)} recipe; torch fp32 CNN on {cores} threads (host has {os.cpu_count()} cores; more "
                     f"threads are slower), effects = single-threaded C restatement (faster than the reference's python "
                     f"per-sample loop, so this baseline is conservative; `reference_shaped` is the SURVEY 8d form)"}
    if reference_shaped:
        out["reference_shaped"] = cpu_reference_shaped(kinds, model, opt, cores)
    return out


def cpu_reference_shaped(kinds, model, opt, cores, batch_cpu: int = 16, n_probe: int = 4410):
    """SURVEY 8d's CPU baseline: B = 16, the flanger / chorus rendered the way the reference renders them -- one python
    iteration of tiny torch ops per sample (oracle.fx.flanger_torch_loop, the execution shape of fx.py:104-115) -- timed on
    `n_probe` samples per effect and extrapolated linearly to 88200 (the loop's cost per sample does not depend on the
    position), plus ONE measured oracle step (phaser in C as pedalboard is, torch fp32 CNN forward / backward / AdamW)."""
    from oracle import fx as ofx, lightning as ol
    from mod_extraction_amd.data_modules import SyntheticFxBatcher
    sampler = SyntheticFxBatcher(batch_cpu, N_SAMPLES, SR, kinds, torch.device("cpu"))
    p = sampler.sample_params()
    src = (torch.rand(batch_cpu, N_SAMPLES + sampler.max_lead) * 2 - 1).mul_(sampler.peak).numpy()
    t0 = time.perf_counter()
    dry, wet, mod = ol.synth_batch(p, sampler.kinds, src, N_SAMPLES, SR, {"flanger": 1.0, "chorus": 30.0})
    t_synth_c = time.perf_counter() - t0
    t0 = time.perf_counter()
    ol.lfo_train_step(model, opt, dry, wet, mod, LOSS)
    t_step = time.perf_counter() - t0
    loop_s, per_sample_us = 0.0, {}
    for kind, max_min_ms in (("flanger", 1.0), ("chorus", 30.0)):
        rows = [i for i, k in enumerate(sampler.kinds) if k == kind]
        if not rows:
            continue
        x = dry[rows, :, :n_probe]
        lfo = torch.rand(len(rows), n_probe)
        par = [torch.as_tensor(p[k])[rows].float() for k in ("feedback", "min_delay_width", "width", "depth", "mix")]
        t0 = time.perf_counter()
        ofx.flanger_torch_loop(x, lfo, ofx.delay_samples(max_min_ms, SR), ofx.delay_samples(10.0, SR), *par)
        dt = time.perf_counter() - t0
        per_sample_us[kind] = round(1e6 * dt / n_probe, 1)
        loop_s += dt * N_SAMPLES / n_probe
    total = t_step + t_synth_c + loop_s
    return {"value": batch_cpu * N_SAMPLES / SR / total, "unit": "audio-seconds/s", "cores": cores, "kind": "port",
            "batch": batch_cpu, "python_loop_us_per_sample": per_sample_us, "seconds": {"cnn_step": round(t_step, 2),
            "effects_in_c": round(t_synth_c, 2), "python_sample_loops_extrapolated": round(loop_s, 2)},
            "sample": f"B = {batch_cpu} x 2 s: one oracle train step (torch fp32 CNN on {cores} threads) + the per-sample python "
                      f"loop of fx.py:104-115 timed on {n_probe} samples per effect module and extrapolated x{N_SAMPLES // n_probe}"}


def conv_key(name, a):
    # (Cin, H) of the launch: positions of Cin / H in each entry point's argument list
    if name == "mx_conv_block_fwd":
        return f"{a[6]}x{a[7]}"
    if name == "mx_conv_block_dgrad":
        return f"64x{a[4]}"
    if name in ("mx_conv_block_fwd_f16", "mx_conv_block_dgrad_f16"):
        return f"64x{a[6]}"
    if name in ("mx_conv_block1_fwd_f16", "mx_conv_block1_wgrad_f16"):
        return f"2x{a[6]}"
    if name == "mx_conv_block1_wgrad_pair_f16":
        return f"2x{a[6]}"            # (Gp, amax, scale, xk_hi, xk_lo, B, H, ...)
    if name in ("mx_conv_block_wgrad_sp_f16", "mx_conv_block_dgrad_sp_f16"):
        return f"64x{a[7]}"          # (.., scale, B, H, Wv, ..)
    if name == "mx_conv_prep_gpool_cl_f16":
        return f"64x{a[4]}"
    if name == "mx_conv_block_wgrad_f16":
        return f"64x{a[6]}"
    if name in ("mx_flanger_fwd", "mx_phaser_fwd"):
        return "fx"
    return f"{a[6]}x{a[7]}"


CONV_NAMES = {"mx_conv_block_fwd", "mx_conv_block_dgrad", "mx_conv_block_wgrad", "mx_conv_block_fwd_f16",
              "mx_conv_block_dgrad_f16", "mx_conv_block_wgrad_f16", "mx_conv_block1_fwd_f16", "mx_conv_block1_wgrad_f16",
              "mx_conv_block_wgrad_sp_f16", "mx_conv_block_dgrad_sp_f16", "mx_conv_prep_gpool_cl_f16",
              "mx_conv_block1_wgrad_pair_f16"}


STREAMING_NAMES = {"mx_conv_prep_gpool_cl_f16"}       # timed next to the convs, but HBM streaming passes


def timed_loop(step, steps, world, device, timer_names, key_fn=None):
    """EXACTLY `steps` steps bracketed by barrier + synchronize on both sides; max over ranks; per-step HIP events.
    N > 1: every gradient all-reduce is bracketed by its own HIP event pair (trainer.CollectiveTimer) and every rank's
    step statistics are gathered AFTER the timed region (`step_ms_per_rank`, `allreduce_ms`): the two numbers that explain
    a scaling-efficiency loss -- time inside the collective (which includes waiting for the slowest rank) and the spread
    of the ranks' own step times."""
    from mod_extraction_amd import _hip, trainer as tr

    def fence():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    marks = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    fence()
    t0 = time.perf_counter()
    with _hip.KernelTimer(timer_names, key_fn) as kt, tr.CollectiveTimer() as ct:
        marks[0].record()
        out = None
        for i in range(steps):
            out = step()
            marks[i + 1].record()
    fence()
    dt = time.perf_counter() - t0
    dt_t = torch.tensor([dt], dtype=torch.float64, device=device)
    if world > 1:
        torch.distributed.all_reduce(dt_t, op=torch.distributed.ReduceOp.MAX)
    per_step = [marks[i].elapsed_time(marks[i + 1]) for i in range(steps)]
    stats = {"min": round(min(per_step), 3), "median": round(statistics.median(per_step), 3), "max": round(max(per_step), 3),
             "slowest_step": per_step.index(max(per_step))}
    if world > 1:
        ar = ct.results_ms()
        mine = {"rank": torch.distributed.get_rank(), "step_ms": stats, "wall_ms_per_step": round(1e3 * dt / steps, 3),
                "allreduce_ms": None if not ar else {"mean": round(mean(ar), 4), "max": round(max(ar), 4),
                                                      "per_step": round(sum(ar) / steps, 4), "calls_per_step": len(ar) / steps}}
        ranks = [None] * world
        try:                                        # (never lose the bench line over the diagnostic gather)
            torch.distributed.all_gather_object(ranks, mine)
        except Exception as e:
            ranks = [mine]
            stats = dict(stats, per_rank_error=repr(e)[:200])
        stats = dict(stats, per_rank=ranks)
    return float(dt_t), kt.results(), stats, out


def scale_fields(step_ms, grad_bytes):
    """N > 1: the scale-readiness numbers of the line, from timed_loop's gathered per-rank statistics."""
    ranks = step_ms.pop("per_rank", None)
    if not ranks:
        return {}
    ar = [r["allreduce_ms"] for r in ranks if r["allreduce_ms"]]
    out = {"step_ms_per_rank": [[r["step_ms"]["min"], r["step_ms"]["median"], r["step_ms"]["max"]] for r in ranks],
           "step_ms_per_rank_fields": "min, median, max of each rank's own steps (HIP events), rank order",
           "rank_median_spread_ms": round(max(r["step_ms"]["median"] for r in ranks) - min(r["step_ms"]["median"] for r in ranks), 3)}
    if ar:
        out["allreduce_ms"] = {"mean": round(mean([a["mean"] for a in ar]), 4), "max": round(max(a["max"] for a in ar), 4),
                               "per_step": round(max(a["per_step"] for a in ar), 4), "calls_per_step": ar[0]["calls_per_step"],
                               "bytes": int(grad_bytes),
                               "note": "HIP events around the flat-gradient sum all-reduce on its stream (includes the wait for the "
                                       "slowest rank); mean / max over calls, per_step = slowest rank's total per step"}
    return out


def fx_floor_pass(batcher, params, n=3):
    """Real and probe-mode (no global traffic inside the loop) durations of the effect launches of one batch."""
    from mod_extraction_amd import _hip
    import contextlib
    res = {}
    for mode in (0, 1):
        with (_hip.probe_twins() if mode else contextlib.nullcontext()), torch.no_grad():
            batcher.render(params)
            with _hip.KernelTimer({"mx_flanger_fwd", "mx_phaser_fwd"}) as kt:
                for _ in range(n):
                    batcher.render(params)
            res[mode] = {k: mean(v) for k, v in kt.results().items()}
    return res


def flanger_lock_steps(batcher, params):
    """Lock-steps of the flanger launch's slowest clip, counted on the host from the integer slot bookkeeping of
    fx.py:95-103 (what csrc/flanger.hip derives on the device): per row of 64 samples the maximal runs without an internal
    read-after-write dependency (a run starting at a ends in front of the first k >= a with k - dep[k] >= a)."""
    import numpy as np
    rows = batcher.rows_fx.cpu().numpy()
    if rows.size == 0:
        return None
    N = batcher.N
    mod = make_mod_host(batcher, params)[rows]                                       # (n_fx, n_lfo)
    up = torch.nn.functional.interpolate(torch.from_numpy(mod).unsqueeze(1), size=N, mode="linear", align_corners=True)[:, 0].numpy()
    M = batcher.max_delay.cpu().numpy()[rows].astype(np.int64)
    ls = (params["width"].float().numpy()[rows] * batcher.max_lfo_delay.cpu().numpy()[rows]).astype(np.float32)
    md = (params["min_delay_width"].float().numpy()[rows] * batcher.max_min_delay.cpu().numpy()[rows]).astype(np.float32)
    worst, total = 0, 0
    k = np.arange(64)[None, :]
    n_rows = -(-N // 64)
    for i in range(rows.size):
        d = (ls[i] * up[i] + md[i]).astype(np.float32)
        w = np.arange(N) % M[i]
        r = np.mod(w.astype(np.float32) - d + np.float32(M[i]), np.float32(M[i]))
        prev = np.clip(np.floor(r).astype(np.int64), 0, M[i] - 1)
        nxt = (prev + 1) % M[i]
        dp = w - prev; dp[dp <= 0] += M[i]
        dn = w - nxt; dn[dn <= 0] += M[i]
        dep = np.minimum(dp, dn)
        dep = np.concatenate([dep, np.full(n_rows * 64 - N, 1 << 30)]).reshape(n_rows, 64)
        t = np.where(dep > k, -1, k - dep)                                           # newest in-row dependency of sample k
        a = np.zeros(n_rows, dtype=np.int64)
        steps = 0
        while True:
            live = a < 64
            if not live.any():
                break
            steps += int(live.sum())
            conflict = (k >= a[:, None]) & (t >= a[:, None])
            first = np.where(conflict.any(axis=1), conflict.argmax(axis=1), 64)
            a = np.where(live, first, a)
        worst, total = max(worst, steps), total + steps
    return {"slowest_clip": worst, "mean": total / rows.size}


def make_mod_host(batcher, params):
    """The LFO rows of a parameter draw on the host (oracle-free: the device LFO synth, copied back)."""
    from mod_extraction_amd.data_modules import make_mod_signals, SHAPE_IDS
    dev = batcher.device
    d = {k: v.to(dev) for k, v in params.items() if isinstance(v, torch.Tensor)}
    shape_id = torch.tensor([SHAPE_IDS[sh] for sh in params["shape"]], dtype=torch.int32, device=dev)
    return make_mod_signals(batcher.n_lfo, batcher.lfo_sr, d["rate_hz"], d["phase"], shape_id, d["exp"]).cpu().numpy()


def lds_roundtrip_ns(device, steps=200000):
    """Time of one dependent LDS round trip of the lock-step's shape (mx_lds_roundtrip_probe), ns."""
    from mod_extraction_amd import _hip
    out = torch.empty(1, device=device)
    _hip.call("mx_lds_roundtrip_probe", steps, _hip.ptr(out), _hip.stream())
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    _hip.call("mx_lds_roundtrip_probe", steps, _hip.ptr(out), _hip.stream())
    b.record()
    torch.cuda.synchronize()
    return 1e6 * a.elapsed_time(b) / steps


def _probe_ns(name, device, steps, *lead_args, per=1, out_floats=1):
    """ns per step of a stand-alone latency microbenchmark entry point (second of two launches timed)."""
    from mod_extraction_amd import _hip
    out = torch.empty(out_floats, device=device)
    _hip.call(name, *lead_args, steps, _hip.ptr(out), _hip.stream())
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    _hip.call(name, *lead_args, steps, _hip.ptr(out), _hip.stream())
    b.record()
    torch.cuda.synchronize()
    return 1e6 * a.elapsed_time(b) / (steps * per)


def lstm_step_ns(device, kind, steps=100000):
    """One bare recurrent step (kind 0 forward, 1 backward) on one 512-lane workgroup: mx_lstm_step_probe, ns."""
    return _probe_ns("mx_lstm_step_probe", device, steps, kind)


def phaser_cascade_ns(device, steps=20000):
    """One (sample, run) of the bare 6-stage all-pass cascade at 8 runs per lane, 512 lanes: mx_phaser_cascade_probe, ns."""
    return _probe_ns("mx_phaser_cascade_probe", device, steps, per=8, out_floats=512)


def fx_blocks(batcher, params, res):
    n_fx, n_ph = int(batcher.rows_fx.numel()), int(batcher.rows_ph.numel())
    N = batcher.N
    out = {}
    if n_fx and "mx_flanger_fwd" in res[0]:
        out["flanger_kernel"] = hbm_block(
            f"flanger_kernel ({n_fx} flanger/chorus clips x {N} samples, fx.py:104-115)", n_fx * N * 8.0,
            res[0]["mx_flanger_fwd"], res[1].get("mx_flanger_fwd"), bound="latency",
            note="8 B/sample: x in, y out; the 882-point LFO is resampled in-kernel. One workgroup (8 producer waves + 1 consumer "
                 "wave) per clip, delay line in LDS; the launch lasts as long as its slowest clip's read-after-write chain, so "
                 "the serial floor, not HBM, is the governing roofline")
        try:                                                # the independent floor: lock-steps x measured LDS round trip
            ls_ = flanger_lock_steps(batcher, params)
            rt = lds_roundtrip_ns(batcher.device)
            fk = out["flanger_kernel"]
            fk["lock_steps"] = ls_
            fk["lds_roundtrip_ns"] = round(rt, 1)
            fk["independent_floor_ms"] = round(ls_["slowest_clip"] * rt * 1e-6, 4)
            fk["frac_of_independent_floor"] = round(fk["independent_floor_ms"] / fk["avg_launch_ms"], 4)
            fk["independent_floor_note"] = ("lock-steps of the slowest clip (host count of the maximal dependency-free runs per row "
                                            "of 64 samples, from fx.py:95-103's integer slot bookkeeping) x the duration of one "
                                            "dependent LDS round trip of the lock-step's shape measured by mx_lds_roundtrip_probe "
                                            "(2 ds_read_b32 -> 5 fp32 ops -> ds_write_b32 on one wavefront, nothing else)")
        except Exception as e:                               # never lose the bench line over the diagnostic
            out["flanger_kernel"]["independent_floor_error"] = repr(e)
    if n_ph and "mx_phaser_fwd" in res[0]:
        lead = params["lead"].double()[batcher.kind_id == 2]
        bytes_ph = float(((lead + N) * (4 + 4 + 2)).sum()) + n_ph * N * 8.0
        out["phaser_kernel"] = hbm_block(
            f"phaser_scan_kernel ({n_ph} clips x ({N} + lead) samples, datasets.py:455-482)", bytes_ph,
            res[0]["mx_phaser_fwd"], res[1].get("mx_phaser_fwd"), bound="latency",
            note="4 B/sample read twice over lead + N samples (both passes of the scan), 8 B/sample written (wet + cropped dry), "
                 "1 B/sample of cut-offs parked and re-read; one workgroup per clip, the clip cut into 512 chunks whose affine "
                 "state maps are built in parallel and chained (csrc/phaser.hip) -- a launch lasts as long as ONE workgroup's "
                 "vector work on its slowest clip (85 of 256 CUs busy), no per-sample dependency chain longer than a chunk")
        try:                                                # the independent floor: bare cascade arithmetic x the scan's run count
            pk = out["phaser_kernel"]
            t_run = phaser_cascade_ns(batcher.device)
            per_lane = -(-int(lead.max() + N) // 512)
            pk["cascade_ns_per_sample_and_run"] = round(t_run, 2)
            pk["independent_floor_ms"] = round(per_lane * 9 * t_run * 1e-6, 4)
            pk["frac_of_independent_floor"] = round(pk["independent_floor_ms"] / pk["avg_launch_ms"], 4)
            pk["independent_floor_note"] = ("samples per lane of the slowest clip (ceil((lead + N) / 512)) x 9 runs (8 of phase A: seven unit "
                                            "states + the driven zero state; 1 of phase C) x the duration of one (sample, run) of the "
                                            "bare 6-stage all-pass cascade + feedback measured by mx_phaser_cascade_probe (8 independent "
                                            "runs per lane on one 512-lane workgroup: no loads, stores, fp64 sin / pow / tan, chunk maps "
                                            "or chaining)")
        except Exception as e:
            out["phaser_kernel"]["independent_floor_error"] = repr(e)
    return out


# ---------------------------------------------------------------------------------------------------------------
def run_lfo_config(args, env, cfg_id):
    from mod_extraction_amd import _hip, trainer as tr
    cfg = CONFIGS[cfg_id]
    rank, world = env["rank"], env["world_size"]
    device = torch.device("cuda", env["local_rank"])
    batch = args.batch or cfg["batch"]
    module, opt, batcher = build_lfo_job(device, rank, batch, cfg["kinds"], overlap=not args.no_overlap)
    if args.conv_precision:
        module.model.conv_precision = args.conv_precision
    runner = tr.Trainer(log_fn=None)

    def step():
        return runner.train_step(module, opt, batcher.next_batch())

    for _ in range(args.warmup):
        step()
    module.logged.clear()
    dt, timings, step_ms, loss = timed_loop(step, args.steps, world, device, CONV_NAMES | {"mx_flanger_fwd", "mx_phaser_fwd"},
                                            conv_key)
    if rank != 0:
        return None
    audio_s = world * batch * (N_SAMPLES / SR) * args.steps
    kernels = {}
    fx_live = {}
    for tag, ms in sorted(timings.items()):
        name, shape = tag.split("#")
        if shape == "fx":
            fx_live[name] = mean(ms)
            continue
        cin, h = (int(v) for v in shape.split("x"))
        blk = BLOCK_H.index(h)
        avg = mean(ms)
        kernels[f"{name[3:]}[block{blk + 1}]"] = {"avg_ms": round(avg, 3)}
        if name not in STREAMING_NAMES:           # (an operand-prep pass has no conv flops to its name)
            kernels[f"{name[3:]}[block{blk + 1}]"]["tflops"] = round(conv_flops(blk, batch) / (avg * 1e-3) / 1e12, 2)
    f16 = "conv_block_fwd_f16[block2]" in kernels
    if f16:                      # roofline kernel = the heaviest conv launch of the step: block-2 forward
        dom = kernels["conv_block_fwd_f16[block2]"]
        roofline = {
            "bound": "mfma", "kernel": "conv_f16x3_dma16_kernel<1, true> (block-2 forward: conv5x13+bias+maxpool on split-fp16 operands, "
                                       "v_mfma_f32_16x16x32_f16)",
            "achieved": dom["tflops"], "peak": F16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": round(dom["tflops"] / F16_MFMA_PEAK_TFLOPS, 4),
            "note": "achieved = ALGORITHMIC fp32-equivalent flops (2*64*64*65*128*345 per clip) / HIP-event launch time; "
                    "each algorithmic MAC group costs 3 fp16 MFMAs (hi*hi + hi*lo + lo*hi), so the matrix pipes execute "
                    "3x that: see achieved_executed / frac_executed; fp32-MFMA peak would be 157.3",
            "achieved_executed": round(3 * dom["tflops"], 1), "frac_executed": round(3 * dom["tflops"] / F16_MFMA_PEAK_TFLOPS, 4),
            "x_fp32_mfma_peak": round(dom["tflops"] / FP32_MFMA_PEAK_TFLOPS, 3),
            "avg_launch_ms": dom["avg_ms"], "flops_per_launch": conv_flops(1, batch),
            "traffic": measured_traffic(batch, "f16"),
            "traffic_unit": "bytes/launch (rocprofv3 PMC pass measured at bs 64, scaled linearly to this batch)"}
    else:
        dom = kernels.get("conv_block_fwd[block2]", {"avg_ms": None, "tflops": None})
        roofline = {
            "bound": "mfma", "kernel": "conv_kernel<1,1,0> (block-2 forward: LayerNorm+conv5x13+bias+maxpool, exact fp32 MFMA)",
            "achieved": dom["tflops"], "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": None if dom["tflops"] is None else round(dom["tflops"] / FP32_MFMA_PEAK_TFLOPS, 4),
            "avg_launch_ms": dom["avg_ms"], "flops_per_launch": conv_flops(1, batch),
            "traffic": measured_traffic(batch, "f32"), "traffic_unit": "bytes/launch (rocprofv3 PMC pass)"}
    mfma_ms = sum(sum(ms) for tag, ms in timings.items() if not tag.endswith("#fx")) / args.steps
    out = {
        "metric": METRIC[cfg_id],
        "value": audio_s / dt, "unit": "audio-seconds/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32",
        "compute_dtype": "f32 tensors; conv MACs as f16x3 products (hi*hi + hi*lo + lo*hi on the fp16 matrix cores, fp32 accumulate)"
                         if f16 else "f32 (exact fp32 MFMA)",
        "data": "synthetic",
        "config": {"workload": f"{cfg['name']}: 2D-CNN LFO extractor train step, bs={batch} x 2 s "
                               f"@44.1 kHz per GPU, {'/'.join(cfg['kinds'])} interleaved, fp32 parity (1e-5)",
                   "baseline_config": cfg_id, "global_batch": world * batch, "n_samples": N_SAMPLES, "parallelism": f"dp{world}",
                   "conv_precision": ("f16x3: every fp32 conv operand is split into an fp16 pair, products = hi*hi + hi*lo + lo*hi "
                                      "on the fp16 matrix cores with fp32 accumulation; fp32-equivalent accuracy (tests/test_gpu_configs.py "
                                      "compares both modes with fp64; all 1e-5 parity tests green); --conv-precision f32 runs exact fp32 MFMA")
                   if f16 else "f32 (exact fp32 MFMA)"},
        "roofline": roofline,
        "kernels": kernels,
        "step_ms": step_ms,
        "conv_ms_per_step": round(mfma_ms, 2),
        "final_loss": None if loss is None else float(loss.detach()),
    }
    out.update(scale_fields(step_ms, opt.flat_grad.numel() * 4))
    # the effect kernels: live durations (side stream, concurrent with the train step) and an isolated pass with the
    # measured serial floor
    with torch.no_grad():
        params = batcher.sample_params()
    torch.cuda.synchronize()
    fxk = fx_blocks(batcher, params, fx_floor_pass(batcher, params))
    for k, v in fxk.items():
        live = fx_live.get("mx_flanger_fwd" if k.startswith("flanger") else "mx_phaser_fwd")
        if live is not None:
            v["avg_launch_ms_in_step"] = round(live, 4)
    out["fx_kernels"] = fxk
    if "flanger_kernel" in fxk:
        out["fx_kernel_frac_of_serial_floor"] = fxk["flanger_kernel"].get("frac_of_serial_floor")
        out["fx_kernel_frac_of_independent_floor"] = fxk["flanger_kernel"].get("frac_of_independent_floor")
        out["fx_kernel_frac_of_hbm"] = fxk["flanger_kernel"]["frac"]
    if world == 1 and f16 and not args.conv_precision and not args.no_fp32_leg:
        module.model.conv_precision = "f32"
        for _ in range(1):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        t = (time.perf_counter() - t0) / 3
        out["exact_fp32_path"] = {"ms_per_step": round(1e3 * t, 2), "value": round(batch * (N_SAMPLES / SR) / t, 1),
                                  "note": "same step with every convolution on v_mfma_f32_32x32x2_f32 (--conv-precision f32), 3 steps"}
    if world == 1 and not args.no_cpu_baseline:
        if args.cpu_baseline_short:         # the launcher's `other_configs` legs: a shorter sample, no reference-shaped leg
            out["cpu_baseline"] = cpu_baseline_lfo(cfg["kinds"], batch_cpu=4, steps=2, reference_shaped=False)
        else:
            out["cpu_baseline"] = cpu_baseline_lfo(cfg["kinds"])
    return out


# ---------------------------------------------------------------------------------------------------------------
def run_config4(args, env):
    import numpy as np
    from mod_extraction_amd import _hip, data_modules, lightning, models, optim
    cfg = CONFIGS[4]
    rank, world = env["rank"], env["world_size"]
    device = torch.device("cuda", env["local_rank"])
    B = args.batch or cfg["batch"]
    W, S = 1024, args.em_step        # train_em_dry_wet.yml: warm-up 1024, steps of 1024 samples
    torch.manual_seed(44); np.random.seed(44)          # same initial weights on every rank, then per-rank data streams
    cnn = models.Spectral2DCNN(**CNN_CFG)
    em = models.LSTMEffectModel()
    # configs/train_em_dry_wet.yml: model_smooth_n_frames 8, should_stretch true, discard_invalid_lfos true.  No pretrained
    # extractor exists in this image and an untrained CNN's output has no valid row at all, so the frozen CNN's forward runs
    # in full and its output is then overwritten with the ground-truth LFO + 1 % noise (what a trained extractor delivers):
    # the validity filter, the row gather and the ragged batch B' <= B are exercised as in a real run.  --em-no-discard
    # gives the round-1/2 form (random extractor output, every row kept).
    class KnownAnswerExtractor(torch.nn.Module):
        def __init__(self, net):
            super().__init__()
            self.net, self.answer = net, None
            self.n_frames = net.n_frames

        def forward(self, x):
            hat, latent = self.net(x)
            if self.answer is not None:
                tgt = util.linear_interpolate_last_dim(self.answer, hat.size(-1), align_corners=True).unsqueeze(1)
                hat = (tgt + 0.01 * (torch.rand_like(tgt) - 0.5)).clamp_(0.0, 1.0) + 0.0 * hat
            return hat, latent

    from mod_extraction_amd import util
    discard = not args.em_no_discard
    extractor = KnownAnswerExtractor(cnn) if discard else cnn
    loss_dict = {"l1": 0.0, "esr": 0.0, "dc": 0.0}
    for item in args.em_loss.split(","):
        name, _, w = item.partition("=")
        loss_dict[name.strip()] = float(w) if w else 1.0
    mod = lightning.TBPTTLFOEffectModeling(W, S, em, lfo_model=extractor, discard_invalid_lfos=discard, should_stretch=True,
                                           model_smooth_n_frames=8, loss_dict=loss_dict).to(device).train()
    opt = optim.FlatAdamW([p for p in mod.parameters() if p.requires_grad], lr=1e-4, betas=(0.8, 0.99))
    sync_replicas(opt)
    torch.manual_seed(44 + rank); np.random.seed(44 + rank)
    bt = data_modules.SyntheticFxBatcher(B, N_SAMPLES, SR, cfg["kinds"], device, audio_seed=44 + rank,
                                         overlap=not args.no_overlap)
    # the batch render and the FROZEN extractor's forward run one batch ahead on the side stream
    def ahead(b):
        if discard:
            extractor.answer = b[2]
        return mod.prepare_ahead((b[0], b[1], None, None))

    bt.ahead_fn = ahead
    kept = []
    n_chunks = (int((338 / 345) * N_SAMPLES) - W) // S
    # the latency-bound recurrence and the prefetch work on disjoint CUs (mod_extraction_amd/streams.py), as in Trainer.fit
    from mod_extraction_amd import streams
    part = None if (args.no_overlap or args.no_cu_partition) else streams.cu_partition(device, main_workgroups=B)
    if part is not None:
        bt.use_side_stream(part[1])
        part[0].wait_stream(torch.cuda.current_stream(device))
        torch.cuda.set_stream(part[0])

    def step():
        dry, wet, _, _ = bt.next_batch()
        out = mod.training_step((dry, wet, None, None), 0, optimizer=opt, world_size=world, prep=bt.last_ahead)
        kept.append(mod.last_kept)
        return out

    for _ in range(max(1, args.warmup)):
        step()
    mod.logged.clear()
    kept.clear()
    names = {"mx_lstm_fwd", "mx_lstm_bwd_l1", "mx_lstm_bwd", "mx_effect_loss_grad", "mx_mrstft_loss", "mx_phaser_fwd", "mx_reduce_rows",
             "mx_adamw_step"}
    dt, timings, step_ms, loss = timed_loop(step, args.steps, world, device, names)
    if rank != 0:
        return None
    # isolated chunk launches with the serial floor
    x = torch.rand(B, 1, S, device=device) * 2 - 1
    lat, wet = torch.rand(B, 1, S, device=device), torch.rand(B, 1, S, device=device) * 2 - 1
    stash, grad = torch.empty(B, S, 384, device=device), torch.zeros(models.LSTM_NPARAM, device=device)
    iso = {}
    import contextlib
    for mode in (0, 1):
        with (_hip.probe_twins() if mode else contextlib.nullcontext()):
            em.clear_hidden()
            y, h0, c0 = em.run_chunk(x, lat, stash)
            em.bptt_l1_chunk(x, lat, y, wet, stash, h0, c0, 1.0 / (B * S), grad)
            with _hip.KernelTimer({"mx_lstm_fwd", "mx_lstm_bwd_l1"}) as kt:
                for _ in range(5):
                    y, h0, c0 = em.run_chunk(x, lat, stash)
                    em.bptt_l1_chunk(x, lat, y, wet, stash, h0, c0, 1.0 / (B * S), grad)
            iso[mode] = {k: mean(v) for k, v in kt.results().items()}
    em.clear_hidden()
    fwd_bytes = B * S * (12.0 + 1536.0)
    bwd_bytes = B * S * (1536.0 + 16.0) + B * 17473 * 4.0
    note = ("one workgroup per clip (512 recurrence lanes; backward + 256 helper lanes); a launch lasts T = 1024 x (one step): LDS "
            "exchange of h / dg + barrier + the dependent FMA / activation chain (csrc/lstm.hip)")
    kernels = {
        "lstm_fwd_kernel": hbm_block(f"lstm_fwd_kernel ({B} clips x {S} steps, models.py:325-339)", fwd_bytes,
                                     iso[0]["mx_lstm_fwd"], iso[1]["mx_lstm_fwd"], bound="latency",
                                     note="12 B/sample I/O + 1536 B/sample BPTT stash written; " + note),
        "lstm_bwd_kernel": hbm_block(
            f"lstm_bwd_kernel ({B} clips x {S} steps, BPTT + L1 + weight gradients, lightning.py:355-384)", bwd_bytes,
            iso[0]["mx_lstm_bwd_l1"], iso[1]["mx_lstm_bwd_l1"], bound="latency",
            note="stash read once (1536 B/sample) + x, lfo, y, wet (16 B/sample) + one gradient row per clip; the weight "
                 "gradients accumulate in the same launch (exact-fp32 MFMA on four helper waves of the workgroup, operands from LDS); " + note),
    }
    # the independent floor: S x one bare dependent step measured by a stand-alone microbenchmark
    for kind, key in ((2, "lstm_fwd_kernel"), (1, "lstm_bwd_kernel")):       # (2: the forward step of the 256-lane kernel the product launches)
        try:
            ns = lstm_step_ns(device, kind)
            k_ = kernels[key]
            k_["step_ns_bare"] = round(ns, 1)
            k_["step_ns_real"] = round(1e6 * k_["avg_launch_ms"] / S, 1)
            k_["independent_floor_ms"] = round(S * ns * 1e-6, 4)
            k_["frac_of_independent_floor"] = round(k_["independent_floor_ms"] / k_["avg_launch_ms"], 4)
            k_["independent_floor_note"] = (
                f"{S} steps x the duration of ONE bare dependent step measured by mx_lstm_step_probe({kind}) on one "
                f"{256 if kind == 2 else 512}-lane workgroup: " +
                                ("LDS broadcast of h -> 32 packed FMAs -> cross-lane add -> v_exp / v_rcp gate -> exchange -> cell "
                                 "update -> tanh -> LDS write -> s_barrier" if kind != 1 else
                                 "gate gradients from LDS -> 16 packed FMAs -> all-reduce over 16 row groups -> dh, dc, dg (local "
                                 "derivatives from seven LDS values) -> LDS write -> s_barrier") +
                "; no global memory, input term, stash, output layer or weight gradients")
        except Exception as e:
            kernels[key]["independent_floor_error"] = repr(e)
    live = {k: mean(v) for k, v in timings.items()}
    per_batch = {k: round(sum(v) / args.steps, 3) for k, v in timings.items()}
    with torch.no_grad():
        params = bt.sample_params()
    torch.cuda.synchronize()
    ahead, bt.ahead_fn = bt.ahead_fn, None
    kernels.update(fx_blocks(bt, params, fx_floor_pass(bt, params)))
    bt.ahead_fn = ahead
    audio_s = world * B * cfg["seconds"] * args.steps
    out = {
        "metric": METRIC[4], "value": audio_s / dt, "unit": "audio-seconds/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"train_em_dry_wet: frozen 2D-CNN LFO extractor + LSTM-64 effect model, truncated BPTT with "
                               f"{W}-sample warm-up and {n_chunks} optimizer steps of {S} samples per batch, L1 loss, bs={B} x 2 s "
                               f"@44.1 kHz per GPU, synthetic dry + this package's phaser render as the wet target",
                   "baseline_config": 4, "global_batch": world * B, "n_samples": N_SAMPLES, "parallelism": f"dp{world}",
                   "optimizer_steps_per_batch": n_chunks, "loss_dict": loss_dict,
                   "lfo_filter": ("model_smooth_n_frames 8, should_stretch, discard_invalid_lfos (train_em_dry_wet.yml); frozen CNN forward "
                                  "runs in full, its output is replaced by the ground-truth LFO + 1 % noise (no pretrained extractor in "
                                  "this image; an untrained one leaves no valid row)") if discard else "none (--em-no-discard)",
                   "clips_trained_per_batch": {"mean": round(mean(kept), 1), "min": min(kept), "of": B},
                   "pipelining": ("batch render + frozen extractor forward of batch i+1 on a side stream under the TBPTT loop of "
                                  "batch i" + ("; recurrence on 20 CUs of every XCD, prefetch work on the other 12 of every XCD (CU-masked streams)" if part is not None
                                               else "")) if not args.no_overlap else "none"},
        "roofline": kernels["lstm_fwd_kernel"],
        "kernels": kernels,
        "ms_per_batch_by_entry_point": per_batch,
        "avg_launch_ms_in_step": {k: round(v, 4) for k, v in live.items()},
        "step_ms": step_ms,
        "final_loss": None if loss is None else float(loss.detach()),
    }
    # reference semantics: `value` counts every clip of the batch (all are rendered and pass the frozen extractor); the
    # LSTM trains on the rows that survive the LFO validity filter -- both rates are quoted
    out["value_clips_rendered"] = out["value"]
    out["value_clips_trained"] = world * mean(kept) * cfg["seconds"] * args.steps / dt
    out.update(scale_fields(step_ms, opt.flat_grad.numel() * 4))
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline_tbptt()
    return out


def cpu_baseline_tbptt(batch_cpu: int = 4):
    import numpy as np
    from oracle import lightning as ol, models as om
    from mod_extraction_amd.data_modules import SyntheticFxBatcher
    cores = min(32, os.cpu_count() or 1)
    torch.set_num_threads(cores)
    torch.manual_seed(44); np.random.seed(44)
    cnn = om.Spectral2DCNN(**CNN_CFG).eval()
    em = om.LSTMEffectModel()
    opt = torch.optim.AdamW(em.parameters(), lr=1e-4, betas=(0.8, 0.99))
    sampler = SyntheticFxBatcher(batch_cpu, N_SAMPLES, SR, ("phaser",), torch.device("cpu"))
    t0 = time.perf_counter()
    p = sampler.sample_params()
    src = (torch.rand(batch_cpu, N_SAMPLES + sampler.max_lead) * 2 - 1).mul_(sampler.peak).numpy()
    dry, wet, _ = ol.synth_batch(p, sampler.kinds, src, N_SAMPLES, SR, {})
    with torch.no_grad():
        hat, _ = cnn(torch.cat([dry, wet], dim=1))
    res = ol.tbptt_common_step(em, opt, dry, wet, hat.squeeze(1), 1024, 1024, {"l1": 1.0, "esr": 0.0, "dc": 0.0},
                               discard_invalid_lfos=False)
    t = time.perf_counter() - t0
    return {"value": batch_cpu * N_SAMPLES / SR / t, "unit": "audio-seconds/s", "cores": cores, "kind": "port",
            "sample": f"ONE batch of {batch_cpu} clips x 2 s through the CPU oracle (C phaser, torch fp32 CNN forward, nn.LSTM TBPTT "
                      f"with {res['steps']} AdamW steps) on {cores} threads"}


# ---------------------------------------------------------------------------------------------------------------
def run_config5(args, env):
    import numpy as np
    from mod_extraction_amd import _hip, data_modules, losses
    cfg = CONFIGS[5]
    rank, world = env["rank"], env["world_size"]
    device = torch.device("cuda", env["local_rank"])
    B, N = args.batch or cfg["batch"], int(cfg["seconds"] * SR)
    torch.manual_seed(45 + rank); np.random.seed(45 + rank)
    bt = data_modules.SyntheticFxBatcher(B, N, SR, cfg["kinds"], device, audio_seed=45 + rank, overlap=not args.no_overlap)
    loss_fn = losses.get_loss_func_by_name("mrstft")

    def step():
        dry, wet, _, _ = bt.next_batch()
        pred = torch.lerp(dry, wet, 0.9).requires_grad_(True)      # stand-in prediction (one pass): the loss path is what is measured
        loss = loss_fn(pred, wet)
        loss.backward()
        return loss.detach()        # (a live graph of the previous step changes the allocator's pattern: one 40 ms hipMalloc)

    for _ in range(args.warmup):
        step()
    dt, timings, step_ms, loss = timed_loop(step, args.steps, world, device, {"mx_flanger_fwd", "mx_mrstft_loss"})
    if rank != 0:
        return None
    live = {k: mean(v) for k, v in timings.items()}
    with torch.no_grad():
        params = bt.sample_params()
    torch.cuda.synchronize()
    kernels = fx_blocks(bt, params, fx_floor_pass(bt, params))
    kernels["flanger_kernel"]["avg_launch_ms_in_step"] = round(live["mx_flanger_fwd"], 4)
    mr = hbm_block("mr_onepass_kernel x {512, 1024, 2048} + mr_fold_all_kernel (losses.py:155-156)",
                   B * N * 12.0, live["mx_mrstft_loss"],
                   note="12 B/sample algorithmic: x, y read once, d loss / d x written once; three STFT resolutions forward and "
                        "backward in one entry point.  One pass per resolution: per frame one forward FFT of x + i y, the loss sums and "
                        "the two linear gradient components G1, G2 (dL/dX = alpha G1 + G2, alpha = the only global scalar); per PAIR "
                        "of frames two inverse FFTs (G1 of both frames, G2 of both), overlap-added in LDS rings; a last gather forms "
                        "alpha g1 + g2.  Nothing per bin reaches memory (the two-pass version parked 12 B per bin: ~25 GB per step, 46x "
                        "the algorithmic bytes; now ~3.5 GB: run sums written and read once).  The kernels are VALU-issue bound "
                        "(FFT butterflies on packed-fp32 instructions), which is why the HBM fraction stays small")
    mr["traffic"] = measured_traffic(B, "mrstft")
    mr["traffic_unit"] = "bytes/launch of mx_mrstft_loss (rocprofv3 PMC passes measured at bs 64, scaled linearly to this batch)"
    if mr["traffic"]:
        mr["traffic_over_algorithmic"] = round(mr["traffic"] / mr["algorithmic_bytes"], 2)
    # the resource that governs these kernels is the vector pipe (FFT butterflies), not HBM: grade them on it, keep the HBM
    # figures as the nominal ones
    flop_clip, per_frame = mrstft_flops_per_clip(N)
    tfl = flop_clip * B / (live["mx_mrstft_loss"] * 1e-3) / 1e12
    mr["nominal_hbm"] = {k: mr[k] for k in ("achieved", "peak", "unit", "frac")}
    mr.update({"bound": "valu", "nominal_bound": "hbm", "achieved": round(tfl, 2), "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
               "frac": round(tfl / FP32_MFMA_PEAK_TFLOPS, 4), "executed_flops_per_launch": flop_clip * B,
               "flop_count": {"per_clip": int(flop_clip), "per_resolution": per_frame, "flop_per_bin": MR_FLOP_PER_BIN,
                              "rule": "per frame: 2 complex FFTs x 5 N log2 N + 40 per bin + 6 N (bench.py:mrstft_flops_per_clip)"},
               "bound_note": "vector-pipe bound: peak = 157.3 TFLOP/s fp32 vector rate (256 CUs x 4 SIMDs x 64 lanes x 2 flop x 2.4 GHz); "
                             "nominal_hbm holds the 12 B/sample HBM figures"})
    pv = measured_valu()
    if pv:
        mr["valu_busy"] = pv.get("valu_busy")                 # SQ_ACTIVE_INST_VALU / SQ_BUSY_CU_CYCLES over the one-pass kernels
        mr["valu_counters"] = pv
    kernels["mrstft_loss"] = mr
    audio_s = world * B * cfg["seconds"] * args.steps
    out = {
        "metric": METRIC[5], "value": audio_s / dt, "unit": "audio-seconds/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"large-batch stress: flanger (fx.py fractional-delay comb) render + multi-resolution STFT loss "
                               f"forward / backward, bs={B} x 4 s @44.1 kHz per GPU (2048 x 4 s over 8 GPUs)",
                   "baseline_config": 5, "global_batch": world * B, "n_samples": N, "parallelism": f"dp{world}"},
        "roofline": mr, "kernels": kernels, "step_ms": step_ms,
        "final_loss": None if loss is None else float(loss.detach()),
    }
    out.update(scale_fields(step_ms, 0))
    if "flanger_kernel" in kernels:
        out["fx_kernel_frac_of_serial_floor"] = kernels["flanger_kernel"].get("frac_of_serial_floor")
        out["fx_kernel_frac_of_independent_floor"] = kernels["flanger_kernel"].get("frac_of_independent_floor")
        out["fx_kernel_frac_of_hbm"] = kernels["flanger_kernel"]["frac"]
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline_stress(N)
    return out


def cpu_baseline_stress(N, batch_cpu: int = 8):
    import numpy as np
    from oracle import lightning as ol, losses as olosses
    from mod_extraction_amd.data_modules import SyntheticFxBatcher
    cores = min(32, os.cpu_count() or 1)
    torch.set_num_threads(cores)
    torch.manual_seed(45); np.random.seed(45)
    sampler = SyntheticFxBatcher(batch_cpu, N, SR, ("flanger",), torch.device("cpu"))
    loss_fn = olosses.get_loss_func_by_name("mrstft")
    t0 = time.perf_counter()
    p = sampler.sample_params()
    src = (torch.rand(batch_cpu, N) * 2 - 1).mul_(sampler.peak).numpy()
    dry, wet, _ = ol.synth_batch(p, sampler.kinds, src, N, SR, {"flanger": 1.0})
    pred = (0.9 * wet + 0.1 * dry).requires_grad_(True)
    loss_fn(pred, wet).backward()
    t = time.perf_counter() - t0
    return {"value": batch_cpu * N / SR / t, "unit": "audio-seconds/s", "cores": cores, "kind": "port",
            "sample": f"ONE step on {batch_cpu} clips x 4 s through the CPU oracle (single-threaded C flanger -- the reference runs a python "
                      f"loop per sample --, torch fp32 MR-STFT forward + backward on {cores} threads)"}


# ---------------------------------------------------------------------------------------------------------------
# output: ONE short JSON line on stdout (the driver parses it; < 6 KB, see tests/test_gpu_step.py), everything else
# in bench_detail.json next to this file and as one `BENCH_DETAIL {...}` line on stderr
LINE_CAP = 6000


def _short(text, n=200):
    if not isinstance(text, str) or len(text) <= n:
        return text
    return text[:n - 3].rstrip() + "..."


def _compact_roofline(r):
    if not r:
        return r
    keep = ("bound", "nominal_bound", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_ms", "flops_per_launch",
            "algorithmic_bytes", "achieved_executed", "frac_executed", "x_fp32_mfma_peak", "serial_floor_ms", "frac_of_serial_floor",
            "independent_floor_ms", "frac_of_independent_floor", "nominal_hbm", "executed_flops_per_launch", "traffic_over_algorithmic", "valu_busy")
    out = {k: r[k] for k in keep if k in r}
    out["kernel"] = _short(r.get("kernel"), 140)
    if r.get("traffic") is not None:
        out["traffic_unit"] = "bytes/launch; committed rocprofv3 PMC pass at bs 64 scaled to this batch, not measured in this run"
    if r.get("note"):
        out["note"] = _short(r["note"], 200)
    return out


def _compact_cpu(c):
    if not c:
        return c
    out = {k: c[k] for k in ("value", "unit", "cores", "kind") if k in c}
    out["value"] = round(out["value"], 3)
    out["sample"] = _short(c.get("sample"), 200)
    if "reference_shaped" in c:
        rs = c["reference_shaped"]
        out["reference_shaped"] = {"value": round(rs["value"], 3), "batch": rs.get("batch"),
                                   "note": "B = 16, flanger / chorus as the reference's per-sample python loop (timed stretch, extrapolated)"}
    return out


def compact(full):
    """The short headline object of a full measurement dict (the LAST stdout line)."""
    out = {k: full[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                "vs_baseline", "dtype", "data") if k in full}
    out["compute_dtype"] = _short(full.get("compute_dtype", full.get("dtype")), 120)
    cfg = full.get("config", {})
    out["config"] = {k: cfg[k] for k in ("baseline_config", "global_batch", "n_samples", "parallelism", "optimizer_steps_per_batch",
                                         "clips_trained_per_batch") if k in cfg}
    out["config"]["workload"] = _short(cfg.get("workload"), 260)
    if "conv_precision" in cfg:
        out["config"]["conv_precision"] = cfg["conv_precision"].split(":")[0].split(" ")[0]
    out["roofline"] = _compact_roofline(full.get("roofline"))
    if "cpu_baseline" in full:
        out["cpu_baseline"] = _compact_cpu(full["cpu_baseline"])
    for k in ("step_ms", "conv_ms_per_step", "fx_kernel_frac_of_serial_floor", "fx_kernel_frac_of_independent_floor",
              "fx_kernel_frac_of_hbm", "value_clips_rendered", "value_clips_trained", "final_loss", "world_size", "dist_backend",
              "rccl_version", "allreduce_ms", "step_ms_per_rank", "rank_median_spread_ms", "worker_rc", "worker_stderr_tail"):
        if k in full:
            out[k] = full[k]
    if "exact_fp32_path" in full:
        out["exact_fp32_path"] = {k: full["exact_fp32_path"][k] for k in ("ms_per_step", "value")}
    for grp in ("kernels", "fx_kernels"):             # floors of the sample-recurrent kernels, numbers only
        for name, k in (full.get(grp) or {}).items():
            if isinstance(k, dict) and "frac_of_independent_floor" in k:
                out.setdefault("recurrent_kernels", {})[name] = {
                    "avg_launch_ms": k.get("avg_launch_ms"), "frac_of_independent_floor": k.get("frac_of_independent_floor"),
                    "frac_of_serial_floor": k.get("frac_of_serial_floor"), "frac_of_hbm": k.get("frac")}
    if "other_configs" in full:
        oc = {}
        for c, o in full["other_configs"].items():
            if "error" in o:
                oc[c] = {"error": o["error"]}
                continue
            r = o.get("roofline") or {}
            oc[c] = {"value": round(o["value"], 1), "ms_per_step": round(o["ms_per_step"], 3),
                     "roofline": {k: r.get(k) for k in ("bound", "frac", "frac_of_independent_floor", "frac_of_serial_floor") if k in r},
                     "fx_kernel_frac_of_independent_floor": o.get("fx_kernel_frac_of_independent_floor"),
                     "cpu_baseline": {"value": round(o["cpu_baseline"]["value"], 3)} if o.get("cpu_baseline") else None}
            if "value_clips_trained" in o:
                oc[c]["value_clips_trained"] = round(o["value_clips_trained"], 1)
        out["other_configs"] = oc
    out["detail"] = "bench_detail.json next to bench.py (also one `BENCH_DETAIL {...}` line on stderr): per-kernel tables, notes, floors"
    line = json.dumps(out)
    if len(line) > LINE_CAP:                          # never let the line outgrow the driver's parser again
        for k in ("step_ms_per_rank", "recurrent_kernels", "exact_fp32_path", "other_configs"):
            if len(line) <= LINE_CAP:
                break
            out.pop(k, None)
            out["dropped_for_length"] = out.get("dropped_for_length", []) + [k]
            line = json.dumps(out)
    return line


def emit(full, detail_path, to_stderr=True):
    """Write the full detail to `detail_path` (and stderr), print the compact line as the LAST thing on stdout."""
    try:
        with open(detail_path, "w") as f:
            json.dump(full, f, indent=1)
    except OSError as e:
        sys.stderr.write(f"bench: could not write {detail_path}: {e}\n")
    if to_stderr:
        sys.stderr.write("BENCH_DETAIL " + json.dumps(full) + "\n")
        sys.stderr.flush()
    sys.stdout.flush()
    print(compact(full), flush=True)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed steps (default: 100 for config 2/3, 5 for 4, 20 for 5)")
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", type=int, default=3, choices=sorted(CONFIGS), help="BASELINE.json configuration (default 3)")
    ap.add_argument("--batch", type=int, default=None, help="clips per GPU (default: the BASELINE config's)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-fp32-leg", action="store_true", help="skip the 3 extra steps on the exact-fp32 convolutions")
    ap.add_argument("--cpu-baseline-short", action="store_true",
                    help="config 2 / 3: a shorter bounded CPU sample (4 clips, 2 steps, no reference-shaped leg)")
    ap.add_argument("--conv-precision", choices=["f16x3", "f32"], default=None,
                    help="arithmetic of the 64->64 convolutions (default: the package default, f16x3)")
    ap.add_argument("--no-cu-partition", action="store_true",
                    help="config 4: let the dispatcher place the recurrence and the prefetch work on the same CUs")
    ap.add_argument("--no-overlap", action="store_true",
                    help="render each batch on the main stream instead of one step ahead on a side stream")
    ap.add_argument("--em-loss", default="l1", help="config 4: comma list name=weight of the effect-model loss "
                    "(default l1=1, the shipped train_em_dry_wet.yml; BASELINE's wording is 'mrstft=1')")
    ap.add_argument("--em-step", type=int, default=1024, help="config 4: samples per truncated-BPTT step (the MR-STFT loss needs "
                    "more than 1024: its 2048-point frames are reflect-padded by 1024 samples, in the reference as well)")
    ap.add_argument("--em-no-discard", action="store_true",
                    help="config 4: no LFO validity filter, random-init extractor output used as is (the round-1/2 form)")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="N = 1, config 3: do not append the configs 2 / 4 / 5 measurements (`other_configs`)")
    ap.add_argument("--detail-out", default=None,
                    help="where the full measurement dict is written (default: bench_detail.json next to bench.py)")
    ap.add_argument("--worker", action="store_true",
                    help="run the measurement in THIS process (set by the launcher; use it under rocprofv3)")
    args = ap.parse_args(argv)
    if args.steps is None:
        args.steps = {2: 100, 3: 100, 4: 5, 5: 20}[args.config]
    return args


# ---------------------------------------------------------------------------------------------------------------
# launcher: `python bench.py --gpus N` outside torchrun.  This process never touches the GPU (it does not even import
# torch); it starts fresh worker processes -- one per rank through torch.distributed.run for N > 1 -- and relays rank
# 0's JSON line.  For the N = 1 headline it then runs the other BASELINE configurations the same way and appends
# them to that line as `other_configs`, so one driver run carries every config.
def _free_port() -> int:
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run_child(cmd, timeout_s, detail_path):
    """Run one worker command; returns (full measurement dict or None, return code, stderr tail, port_in_use).  The worker
    writes its full dict to `detail_path` (its stdout carries only the compact line); its stderr is relayed without the
    BENCH_DETAIL line (the launcher prints ONE merged detail line itself).  `port_in_use` is looked for in the WHOLE
    stderr (torchrun's multi-rank tracebacks are longer than the tail)."""
    import subprocess
    if os.path.exists(detail_path):
        os.remove(detail_path)
    try:
        res = subprocess.run(cmd + ["--detail-out", detail_path], cwd=ROOT, capture_output=True, text=True, timeout=timeout_s)
    except subprocess.TimeoutExpired as e:
        return None, 124, f"timeout after {timeout_s} s: {' '.join(cmd)}\n{(e.stderr or '')[-2000:]}", False
    err = "\n".join(ln for ln in res.stderr.splitlines() if "BENCH_DETAIL " not in ln)
    if err:
        sys.stderr.write(err + "\n")
        sys.stderr.flush()
    out = None
    try:
        with open(detail_path) as f:
            out = json.load(f)
        os.remove(detail_path)
    except (OSError, ValueError):
        out = None
    port_in_use = "EADDRINUSE" in res.stderr or "address already in use" in res.stderr.lower()
    return out, res.returncode, err[-2000:], port_in_use


def _worker_cmd(args, config, steps, extra=()):
    cmd = ["--gpus", str(args.gpus), "--config", str(config), "--steps", str(steps), "--warmup", str(args.warmup)]
    if args.batch:
        cmd += ["--batch", str(args.batch)]
    if args.conv_precision:
        cmd += ["--conv-precision", args.conv_precision]
    for flag in ("no_cu_partition", "no_overlap"):
        if getattr(args, flag):
            cmd.append("--" + flag.replace("_", "-"))
    if args.em_loss != "l1":
        cmd += ["--em-loss", args.em_loss]
    if args.em_no_discard:
        cmd.append("--em-no-discard")
    if args.em_step != 1024:
        cmd += ["--em-step", str(args.em_step)]
    return cmd + list(extra)


def launch(args) -> int:
    script = os.path.join(ROOT, "bench.py")
    import tempfile
    tmp_detail = os.path.join(tempfile.gettempdir(), f"bench_detail_{os.getpid()}.json")
    flags = (["--no-cpu-baseline"] if args.no_cpu_baseline else []) + (["--no-fp32-leg"] if args.no_fp32_leg else [])
    for _ in range(4):                          # (a free port can be taken between finding it and torchrun's bind: try another one)
        if args.gpus > 1:
            cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
                   "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), script] + \
                  _worker_cmd(args, args.config, args.steps, flags)
        else:
            cmd = [sys.executable, script, "--worker"] + _worker_cmd(args, args.config, args.steps, flags)
        out, rc, err, port_in_use = _run_child(cmd, 3000, tmp_detail)
        if out is not None or args.gpus == 1 or not port_in_use:     # (a rendezvous that fails on the port fails within seconds)
            break
    if out is None:
        sys.stderr.write(f"bench worker failed (rc {rc})\n")
        return rc or 1
    # a worker that printed its line and THEN crashed (e.g. at process teardown) is still a failed run: the code is on
    # record in the line and becomes the launcher's exit code
    out["worker_rc"] = rc
    if rc != 0:
        out["worker_stderr_tail"] = err[-400:]
    worst = rc
    if args.gpus == 1 and args.config == 3 and not args.no_other_configs and not args.batch and not args.conv_precision:
        others = {}
        for c, steps in ((2, 30), (4, 5), (5, 20)):
            t0 = time.perf_counter()
            # every config carries its own bounded cpu_baseline: short samples, ~5-10 s each
            leg = ["--no-fp32-leg"] + (["--no-cpu-baseline"] if args.no_cpu_baseline else ["--cpu-baseline-short"])
            o, rc_c, err_c, _ = _run_child([sys.executable, script, "--worker"] + _worker_cmd(args, c, steps, leg), 900, tmp_detail)
            worst = worst or rc_c
            if o is None:
                others[str(c)] = {"error": f"rc {rc_c}", "stderr_tail": err_c[-400:]}
                continue
            keep = {k: v for k, v in o.items() if k not in ("n_gpus", "higher_is_better", "scaling", "vs_baseline", "data", "config",
                                                            "world_size", "dist_backend", "rccl_version")}
            keep["config"] = o["config"]
            keep["process_wall_s"] = round(time.perf_counter() - t0, 1)
            keep["worker_rc"] = rc_c
            if rc_c != 0:
                keep["worker_stderr_tail"] = err_c[-400:]
            others[str(c)] = keep
        out["other_configs"] = others
    emit(out, args.detail_out or os.path.join(ROOT, "bench_detail.json"))
    return worst


def main():
    args = parse_args()
    under_torchrun = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    if not (args.worker or under_torchrun):
        sys.exit(launch(args))

    global torch
    import torch as _torch
    torch = _torch
    from mod_extraction_amd import trainer as tr
    env = tr.init_distributed()
    world = env["world_size"]
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    torch.cuda.set_device(torch.device("cuda", env["local_rank"]))
    if args.config in (2, 3):
        out = run_lfo_config(args, env, args.config)
    elif args.config == 4:
        out = run_config4(args, env)
    else:
        out = run_config5(args, env)
    if env["rank"] == 0:
        # how many ranks the collective library actually saw (a SCALE record must show it)
        out["world_size"] = torch.distributed.get_world_size() if torch.distributed.is_initialized() else 1
        out["dist_backend"] = torch.distributed.get_backend() if torch.distributed.is_initialized() else None
        try:
            out["rccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception:
            out["rccl_version"] = None
        emit(out, args.detail_out or os.path.join(ROOT, "bench_detail.json"))
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
