#!/usr/bin/env python3
"""bench.py — headline benchmark of the MI355X path-tracing hot path.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one full pass of the hot path over the headline workload (BASELINE.json configs[2]):
Cornell box (scenes/cornellBoxSetting.json, unchanged), 1920x1080, 1024 spp
(superSamples 4 x samples 64), repaired (L1) semantics, max 8 bounces, fp64, seed 0x5EED, sin/cos as the
host libm returns them (RTM_MODE_HOST_TRIG; --device-trig is the labelled ~2 % faster row), rendered into the
HBM-resident float3 accumulation buffer.  Kernel: since round 4 the fp64 TOLERANCE row (variant 18: the default kernel's
source with FMA contraction and one-ulp division / square root; asserted within north_star's 1e-4 per pixel, observed 0
differing pixels on every BASELINE Cornell configuration — tests/test_tolerance_gpu.py); the bit-exact
default kernel's row is measured beside it in the same run (other_configs.headline_frame_bit_exact_kernel), and --exact
makes it the headline.  With N > 1 the image is dealt out in interleaved
8-row bands (band b -> rank b mod N; total work fixed => "strong"), and one RCCL gather to rank 0
ends every step inside the timed region.  metric = Msamples/s = W*H*spp / s, whole job.

Extra objects on the JSON line:
  roofline     — the binding roof of the render kernel is the fp64 vector ALU, so
                 achieved = algorithmic flops per launch / average kernel time (HIP events on the
                 launch stream); an "hbm" sub-object carries the achieved HBM GB/s the metric asks
                 for (algorithmic bytes: the float3 image written once + the scene read).
  cpu_baseline — the CPU oracle (oracle/cpu_ref.c, OpenMP, all host cores) on a bounded sample of
                 the same workload (rank 0, N = 1 only).
  with_d2h     — the same steps with the frame copied to pinned host memory inside the timed region
                 (SURVEY.md §8d wall time: kernel + final D2H); never the headline value.
  other_configs— the bit-exact kernel's headline row, BASELINE configs[1], configs[4] through the grid AND through the
                 exhaustive pipeline (whole frame), the plane scene and the labelled rows, measured in this run outside the
                 timed region (N = 1 only); every row carries its own roofline object (roofline_of: the work model follows
                 the kernel the library reports, never a frac above 1).
  N > 1        — frame_matches_single_gpu, config.per_rank, gather_ms (distributed.multi_gpu_evidence): the line's own proof
                 that N ranks rendered the right frame, all outside the timed region.
Every field says whether it was measured in this run; numbers replayed from committed profiles carry
"measured_in_run": false and the file they come from.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HEADLINE = dict(scene="cornellBoxSetting.json", width=1920, height=1080, samples=64, super_samples=4,
                mode="repaired", max_bounces=8, seed=0x5EED)
PEAK_FP32_VECTOR_TFLOPS = 157.3  # MI355X_MICROARCH.md chip table (packed fp32, FMA = 2 flops)
PEAK_FP64_VECTOR_TFLOPS = 78.6   # AMD datasheet (FMA = 2 flops); MI355X_MICROARCH.md has fp32 157.3
PEAK_HBM_GBS = 8000.0            # MI355X_MICROARCH.md chip table (spec)


def algorithmic_flops_per_sample(n_spheres, casts_per_sample, bounces_per_sample, d4_fraction):
    """F_sample = 50 + C*F_cast + B*F_bounce; F_cast = 17 N + 3 N_{D4>=0} + 18;
    F_bounce = 83.  C, B measured by the kernel's counters, N_{D4>=0} by the oracle's."""
    f_cast = 17.0 * n_spheres + 3.0 * n_spheres * d4_fraction + 18.0
    return 50.0 + casts_per_sample * f_cast + bounces_per_sample * 83.0


def cpu_baseline(cfg, budget_rows):
    """Time the CPU oracle on `budget_rows` full-width rows of the headline workload."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _oracle
    st, arr, n = _oracle.load_scene(_oracle.scene_path(cfg["scene"]), width=cfg["width"],
                                    height=cfg["height"], samples=cfg["samples"],
                                    super_samples=cfg["super_samples"])
    threads = _oracle.lib().rtmo_max_threads()
    r0 = cfg["height"] // 2 - budget_rows // 2
    opt = _oracle.make_options(mode=1, max_bounces=cfg["max_bounces"], seed=cfg["seed"],
                               row_begin=r0, row_end=r0 + budget_rows)
    # warm the thread pool on one row
    _oracle.render(st, arr, n, _oracle.make_options(mode=1, max_bounces=cfg["max_bounces"],
                                                    seed=cfg["seed"], row_begin=0, row_end=1))
    t0 = time.perf_counter()
    _, cnt = _oracle.render(st, arr, n, opt, threads=threads, structure=0)
    dt = time.perf_counter() - t0
    t1 = time.perf_counter()
    _oracle.render(st, arr, n, _oracle.make_options(mode=1, max_bounces=cfg["max_bounces"],
                                                    seed=cfg["seed"], row_begin=r0,
                                                    row_end=r0 + max(1, budget_rows // 8)),
                   threads=threads, structure=1, want_counters=False)
    dt_ref = time.perf_counter() - t1
    samples_ref = max(1, budget_rows // 8) * cfg["width"] * cfg["samples"] * cfg["super_samples"] ** 2
    t2 = time.perf_counter()
    _, c1 = _oracle.render(st, arr, n, _oracle.make_options(mode=1, max_bounces=cfg["max_bounces"], seed=cfg["seed"],
                                                            row_begin=r0, row_end=r0 + 1), threads=1, structure=0)
    dt_1 = time.perf_counter() - t2
    return {
        "value": cnt["samples"] / dt / 1e6, "unit": "Msamples/s", "cores": threads, "kind": "port",
        "sample": f"rows {r0}..{r0 + budget_rows} of the headline frame at full 1024 spp "
                  f"({cnt['samples']} samples, {dt:.2f} s wall, OpenMP per-pixel parallel)",
        "reference_loop_structure_value": samples_ref / dt_ref / 1e6,
        "single_thread_value": c1["samples"] / dt_1 / 1e6,
        "casts_per_sample": cnt["casts"] / cnt["samples"],
        "d4_fraction": cnt["sphere_tests_d4"] / max(1, cnt["sphere_tests"]),
    }


def measured_fp64_peak(rtm):
    """"FP64 vector 78.6 TF (AMD datasheet; verify by microbenchmark)".  A chip-filling v_fma_f64
    kernel (rtm_debug_fp64_peak: 8 independent accumulators per lane, 4 and 8 waves per SIMD, >= 60 ms per launch so
    that the clock has settled), HIP-event time, FMA = 2 flops.  The larger of the two is the measured peak."""
    import ctypes as C
    rows = {}
    for waves in (4, 8):
        tf, ms = C.c_double(), C.c_double()
        rtm._lib.check(rtm.lib().rtm_debug_fp64_peak(waves, 60.0, C.byref(tf), C.byref(ms)), "rtm_debug_fp64_peak")
        rows[f"{waves}_waves_per_simd"] = {"tflops": tf.value, "kernel_ms": ms.value}
    best = max(v["tflops"] for v in rows.values())
    return {"value": best, "unit": "TFLOP/s", "measured_in_run": True, "rows": rows,
            "implied_clock_ghz_at_4_cycles_per_wave_instruction": best * 1e12 / (2 * 64 * 1024) * 4 / 1e9,
            "how": "chip-filling v_fma_f64 kernel, HIP events, FMA = 2 flops (rtm_debug_fp64_peak; the per-instruction "
                   "price list of the same method: profiles/r3/fp64_peak.txt)"}


KERNEL_OF_VARIANT = {1: "render_tiles_kernel (per-object loop, compiler math)", 2: "render_tiles_kernel", 3: "render_tiles_kernel",
                     7: "render_tiles_kernel (stamped)", 9: "render_tiles_kernel", 12: "wf_nearest_f32_kernel + wf_shade_kernel",
                     14: "render_tiles_kernel", 15: "render_tiles_kernel (primary-hit reuse)", 16: "render_fp32_kernel",
                     17: "render_grid_kernel + grid_finalize_kernel", 18: "rtm_tol::render_tiles_kernel (+ prim_prepass_kernel)"}


def roofline_of(st, kernel_ms, n_objects, width, d4_fraction=None, peak_measured=None, replay=None):
    """The roofline object of ONE measured row, from the library's own account of what ran (rtm_stats.variant) — never from
    the scene's size.  st: the stats of an instrumented step of the row; kernel_ms: its render kernels' time.
      exhaustive fp64 kernels (1, 2, 3, 9, 14, 15, 18)  the reference's flops per sample against the fp64
                                                        vector peak;
      12, the exhaustive large-scene pipeline           16 flops (8 packed-fp32 FMAs) per (ray, sphere) pair against the
                                                        packed-fp32 peak;
      16, the labelled fp32 row                         the reference's flops against the fp32 peak;
      17, the uniform grid                              NO flop roofline: the kernel returns the reference loop's answer
                                                        from a different algorithm, so pricing it at the reference's 17 N
                                                        flops per cast would report work it does not do.  It gets the
                                                        HBM view the contract defines (algorithmic bytes / time), its
                                                        measured sphere tests per cast, the factor by which that undercuts
                                                        the reference's loop, and (replayed from the committed profile)
                                                        VALU busy x active lanes and the counter traffic.
    frac is never above 1."""
    variant = st["variant"]
    samples, casts, bounces = st["samples"], st["casts"], st["bounces"]
    cps, bps = casts / samples, bounces / samples
    t = kernel_ms * 1e-3
    out = {"kernel": KERNEL_OF_VARIANT.get(variant, f"variant {variant}"), "kernel_ms": kernel_ms, "variant": variant,
           "casts_per_sample": cps, "bounces_per_sample": bps, "measured_in_run": True}
    alg_bytes = st["pixels"] * 12 + n_objects * 96 if st.get("pixels") else None
    if variant == 17:
        grid_bytes = (replay or {}).get("grid_bytes")
        ab = (alg_bytes or 0) + (grid_bytes or 0)
        ach = ab / t / 1e9
        out.update({"bound": "hbm", "achieved": ach, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": min(1.0, ach / PEAK_HBM_GBS),
                    "algorithmic_bytes_per_launch": ab,
                    "traffic": (replay or {}).get("traffic"), "traffic_source": (replay or {}).get("traffic_source"),
                    "flop_roofline": None,
                    "reference_object_tests_per_cast": n_objects,
                    "reference_flops_per_cast": 17.0 * n_objects + 18.0,
                    "note": "the nearest hit of the reference's loop over all objects (src/Renderer.cpp:58-73) through a uniform "
                            "grid: same hit object, distance and image from a fraction of the Intersect calls — a different "
                            "algorithm for the same answer, so there is no flop roofline on the reference's work model (priced at "
                            "17 N flops per cast the row would read thousands of TFLOP/s).  Bound by per-lane gathers and lane "
                            "divergence, not by HBM: the HBM view is the contract's, reported, and tiny."})
        if st.get("object_tests"):
            tpc = st["object_tests"] / casts
            out.update({"object_tests_per_cast": tpc, "object_tests_source": "rtm_stats.object_tests of a counting render of the same "
                        "frame in this run (RTM_MODE_COUNT_TESTS)", "undercuts_reference_tests_by": n_objects / tpc,
                        "object_tests_per_s": st["object_tests"] / t})
        if replay and replay.get("valu_issue"):
            out["valu_issue"] = replay["valu_issue"]
        return out
    if variant == 12:
        tests_per_s = casts * n_objects / t
        ach = tests_per_s * 16.0 / 1e12
        out.update({"bound": "valu-fp32", "achieved": ach, "peak": PEAK_FP32_VECTOR_TFLOPS, "unit": "TFLOP/s",
                    "frac": min(1.0, ach / PEAK_FP32_VECTOR_TFLOPS), "object_tests_per_s": tests_per_s,
                    "traffic": (replay or {}).get("traffic"), "traffic_source": (replay or {}).get("traffic_source"),
                    "note": "brute force over every sphere; a pair whose discriminant is provably negative is rejected by 8 "
                            "packed-fp32 FMAs (16 flops, counted here), the rest get the reference's fp64 arithmetic"})
        return out
    d4 = 0.0 if d4_fraction is None else d4_fraction
    f_sample = algorithmic_flops_per_sample(n_objects, cps, bps, d4)
    ach = f_sample * samples / t / 1e12
    peak = PEAK_FP32_VECTOR_TFLOPS if variant == 16 else PEAK_FP64_VECTOR_TFLOPS
    out.update({"bound": "valu-fp32" if variant == 16 else "valu-fp64", "achieved": ach, "peak": peak, "unit": "TFLOP/s",
                "frac": min(1.0, ach / peak), "flops_per_sample": f_sample,
                "d4_fraction": d4_fraction if d4_fraction is not None else "not measured for this scene: the 3 N_{D4>=0} term is left out",
                "traffic": (replay or {}).get("traffic"), "traffic_source": (replay or {}).get("traffic_source"),
                "hbm": ({"achieved": alg_bytes / t / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": alg_bytes / t / 1e9 / PEAK_HBM_GBS,
                         "algorithmic_bytes_per_launch": alg_bytes} if alg_bytes else None)})
    if peak_measured and variant != 16:
        out["frac_of_measured_peak"] = min(1.0, ach / peak_measured)
    return out


# Issue cost of a wave64 VALU instruction per class, in cycles of the SIMD it occupies, from the wall-clock price list
# of profiles/ubench/fp64_peak.hip (profiles/r3/fp64_peak.txt: ns per wave-instruction per SIMD relative to v_fma_f64 = 4
# cycles: fp64 add/mul/fma/min/ldexp/compare, v_cndmask_b32 with an SGPR mask, conversions, 64-bit moves, v_mul_lo/hi_u32
# and v_mad_u64_u32 all cost the same 4; v_mov_b32, v_add_u32, v_xor_b32 and v_fma_f32 cost 2; v_rcp/v_rsq_f64 13.9;
# v_sqrt_f32 7).  INT32 and "other" are mixes: weights from the static histogram of the hot path
# (profiles/r2/isa_hotpath_histogram.txt): INT32 = 13 multiplies (4) + 112 add/xor/shift/compare (2) per trip; other =
# 51 selects + 25 fp64 compares + 12 fix-up/ldexp (4) + 21 moves, mostly 64-bit (4) + 5 lane ops (4).
VALU_CYCLES = {"SQ_INSTS_VALU_ADD_F64": 4.0, "SQ_INSTS_VALU_MUL_F64": 4.0, "SQ_INSTS_VALU_FMA_F64": 4.0,
               "SQ_INSTS_VALU_TRANS_F64": 13.9, "SQ_INSTS_VALU_INT32": (13 * 4.0 + 112 * 2.0) / 125, "SQ_INSTS_VALU_INT64": 4.0,
               "SQ_INSTS_VALU_CVT": 4.0, "SQ_INSTS_VALU_FMA_F32": 2.0, "SQ_INSTS_VALU_MUL_F32": 2.0, "SQ_INSTS_VALU_ADD_F32": 2.0,
               "SQ_INSTS_VALU_TRANS_F32": 7.0, "other": 4.0}


def valu_issue_view(pmc, source):
    """Instruction-issue view of the render kernel from a committed PMC pass of the same command: the hardware's own
    busy figure (rocprof's derived VALUBusy = SQ_ACTIVE_INST_VALU x 4 / SIMDs / GRBM_GUI_ACTIVE: the SQ cycle counters
    tick in units of four cycles) and the same from the dynamic mix priced per class with VALU_CYCLES."""
    cyc = pmc["GRBM_GUI_ACTIVE"] / 8.0  # counter is summed over the 8 XCDs
    total = pmc["SQ_INSTS_VALU"]
    classes = {k: pmc.get(k, 0.0) for k in VALU_CYCLES if k != "other"}
    classes["other"] = total - sum(classes.values())
    priced_cycles = sum(classes[k] * VALU_CYCLES[k] for k in classes)
    return {"wave_instructions_per_launch": total, "simds": 1024, "kernel_cycles": cyc,
            "clock_ghz": cyc / (pmc["kernel_stats"][0]["AverageNs"] if isinstance(pmc["kernel_stats"][0]["AverageNs"], float)
                                else float(pmc["kernel_stats"][0]["AverageNs"])),
            "valu_busy_hw": pmc["SQ_ACTIVE_INST_VALU"] * 4.0 / 1024.0 / cyc,
            "valu_busy_priced": priced_cycles / 1024.0 / cyc,
            "priced_cycles_per_instruction": priced_cycles / total,
            "class_counts": classes, "class_cycles": VALU_CYCLES,
            "fp64_arith_share": (classes["SQ_INSTS_VALU_ADD_F64"] + classes["SQ_INSTS_VALU_MUL_F64"] +
                                 classes["SQ_INSTS_VALU_FMA_F64"] + classes["SQ_INSTS_VALU_TRANS_F64"]) / total,
            "active_lanes_frac": pmc.get("SQ_THREAD_CYCLES_VALU", 0.0) / total / 64.0,
            "wave_wait_share": {"SQ_WAIT_INST_ANY_over_SQ_WAVE_CYCLES": pmc.get("SQ_WAIT_INST_ANY", 0.0) / max(1.0, pmc.get("SQ_WAVE_CYCLES", 1.0)),
                                "resident_waves_per_simd": pmc.get("SQ_WAVE_CYCLES", 0.0) * 4.0 / 1024.0 / cyc,
                                "note": "4 resident waves share one VALU that is busy ~all the time, so each wave issues at most a "
                                        "quarter of the time; about half of a wave's resident time is spent waiting on its own "
                                        "previous instruction (dependent fp64 chains), the rest ready but not picked"},
            "measured_in_run": False, "source": source}


def replayed_profile(name):
    """PMC-derived figures of a committed rocprofv3 pass of the same command (profiles/prof_*.sh), newest round first."""
    for rnd in ("r4", "r3"):
        path = os.path.join(ROOT, "profiles", rnd, name)
        if os.path.exists(path):
            return json.load(open(path)), f"profiles/{rnd}/{name}"
    return None, None


def grid_replay():
    gj, src = replayed_profile("c5_grid_pmc_summary.json")
    if not gj:
        return None
    out = {}
    if "FETCH_SIZE" in gj and "WRITE_SIZE" in gj:
        out["traffic"] = 1024.0 * sum(gj.get(k, 0.0) for k in ("FETCH_SIZE", "WRITE_SIZE", "grid_finalize.FETCH_SIZE",
                                                                 "grid_finalize.WRITE_SIZE"))
        out["traffic_source"] = {"measured_in_run": False, "file": src,
                                 "note": "FETCH_SIZE + WRITE_SIZE per frame, render_grid_kernel + grid_finalize_kernel (rocprofv3 "
                                         "--pmc, separate passes; profiles/prof_c5_grid.sh)"}
    if "SQ_INSTS_VALU" in gj and "GRBM_GUI_ACTIVE" in gj:
        cyc = gj["GRBM_GUI_ACTIVE"] / 8.0
        busy = gj.get("SQ_ACTIVE_INST_VALU", 0.0) * 4.0 / 1024.0 / cyc
        lanes = gj.get("SQ_THREAD_CYCLES_VALU", 0.0) / gj["SQ_INSTS_VALU"] / 64.0
        out["valu_issue"] = {"valu_busy_hw": busy, "active_lanes_frac": lanes, "busy_x_active_lanes": busy * lanes,
                             "wave_instructions_per_launch": gj["SQ_INSTS_VALU"], "measured_in_run": False, "source": src,
                             "note": "the share of the VALU's lane-slots that do work: what bounds this kernel (divergent walks, "
                                     "per-lane gathers), next to an L2 hit rate of " + (f"{gj['l2_hit_rate']:.2f}" if "l2_hit_rate" in gj else "?")}
    out["grid_bytes"] = gj.get("grid_bytes", 22.0e6)
    return out


def other_configs(rtm, cfg, device, host_trig, full_c5=True, cpu_rows=64, d4_cornell=None, peak_measured=None, headline_variant=0):
    """BASELINE configs[1], configs[4] (the full frame through the grid kernel AND through the exhaustive pipeline), the
    plane scene and the labelled rows, measured in this run, outside the timed region.  EVERY row carries its own roofline
    object (roofline_of: the work model follows the kernel the library reports).  Every group of rows stands alone: a
    failure is recorded under its name and the others are still measured."""
    import torch
    out = {"measured_in_run": True}

    def timed(r, steps, **kw):
        r.render_rows_device(want=("f32",), stats=True, **kw)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            r.render_rows_device(want=("f32",), stats=False, **kw)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        o, st = r.render_rows_device(want=("f32",), stats=True, **kw)
        st["pixels"] = int(o["f32"].shape[0]) * int(o["f32"].shape[1])
        return dt, st

    def row_of(dt, st, steps, n_objects, width, d4=None, replay=None, **extra):
        row = {"value": st["samples"] / dt / 1e6, "unit": "Msamples/s", "ms_per_step": dt * 1e3, "steps": steps,
               "kernel_ms": st["kernel_ms"], "casts_per_sample": st["casts"] / st["samples"],
               "variant": rtm.lib().rtm_variant_name(st["variant"]).decode(), "split": st["split"],
               "roofline": roofline_of(st, st["kernel_ms"], n_objects, width, d4, peak_measured, replay)}
        row.update(extra)
        return row

    def guarded(name, fn):
        try:
            fn()
        except Exception as exc:  # noqa: BLE001 — the bench line must come out; the failure is named in it
            out[name] = {"error": repr(exc)}

    def c2():
        data = rtm.LoadData(os.path.join(ROOT, "scenes", "cornellBoxSetting.json")).data
        data.width, data.height, data.samples, data.superSamples = 512, 512, 16, 4
        for name, mb in (("c2_cornell_512x512_256spp_max8", 8), ("c2_cornell_512x512_256spp_unlimited", -1)):
            r = rtm.Renderer(data, mode="repaired", max_bounces=mb, seed=cfg["seed"], device=device, host_trig=host_trig)
            dt, st = timed(r, 5)
            out[name] = row_of(dt, st, 5, len(data.object), 512, d4_cornell)
            # ... and the same frame through the labelled fp64 tolerance row (variant 18; any depth since round 4's last step)
            rt = rtm.Renderer(data, mode="repaired", max_bounces=mb, seed=cfg["seed"], device=device, host_trig=host_trig, variant=18)
            dt, st = timed(rt, 5)
            out["LABELLED_" + name + "_fp64_tolerance"] = row_of(dt, st, 5, len(data.object), 512, d4_cornell)

    def headline_data():
        data = rtm.LoadData(os.path.join(ROOT, "scenes", "cornellBoxSetting.json")).data
        data.width, data.height, data.samples, data.superSamples = cfg["width"], cfg["height"], cfg["samples"], cfg["super_samples"]
        return data

    def reuse_row():
        # SEPARATELY LABELLED row: the headline frame with the primary hit of a sub-pixel computed once
        # for its S samples (variant 15) — same image and counters, less work per sample than the reference does
        data = headline_data()
        r = rtm.Renderer(data, mode="repaired", max_bounces=cfg["max_bounces"], seed=cfg["seed"], device=device,
                         host_trig=host_trig, variant=15)
        dt, st = timed(r, 3)
        out["LABELLED_headline_frame_with_primary_hit_reuse"] = row_of(
            dt, st, 3, len(data.object), cfg["width"], d4_cornell,
            note="not comparable with the headline value or the CPU baseline: one nearest-hit search per sub-pixel instead of one "
                 "per sample for the primary ray (the reference repeats it, src/Renderer.cpp:224-238); its roofline prices the "
                 "reference's flops, which this row does not all execute")

    def tolerance_row():
        # SEPARATELY LABELLED row: the default kernel's source compiled with FMA contraction and one-ulp division / square
        # root (variant 18, csrc/rtm_kernels_tol.hip) — north_star's tolerance is 1e-4 per pixel, the default kernels meet
        # it with 0.  Reported with its pixel differences against the exact frame of this run and its own roofline.
        data = headline_data()
        exact, est = rtm.Renderer(data, mode="repaired", max_bounces=cfg["max_bounces"], seed=cfg["seed"], device=device,
                                  host_trig=host_trig).render_rows_device(want=("f64",), stats=True)
        r = rtm.Renderer(data, mode="repaired", max_bounces=cfg["max_bounces"], seed=cfg["seed"], device=device,
                         host_trig=host_trig, variant=18)
        tol, _ = r.render_rows_device(want=("f64",), stats=True)
        a, b = exact["f64"], tol["f64"]
        delta = (a - b).abs()
        differing = int((a.view(torch.int64) != b.view(torch.int64)).any(dim=2).sum())
        outside = int((delta.amax(dim=2) > 1e-4).sum())
        del exact, tol
        if headline_variant == 18:
            # the headline of this line IS the tolerance row: the bit-exact default kernel's row stands beside it
            rx = rtm.Renderer(data, mode="repaired", max_bounces=cfg["max_bounces"], seed=cfg["seed"], device=device,
                              host_trig=host_trig)
            dtx, stx = timed(rx, 5)
            xj, xsrc = replayed_profile("default_pmc_summary.json")
            out["headline_frame_bit_exact_kernel"] = row_of(
                dtx, stx, 5, len(data.object), cfg["width"], d4_cornell,
                executed_valu_wave_instructions=({"value": xj["SQ_INSTS_VALU"], "measured_in_run": False, "source": xsrc} if xj else None),
                note="variant 0, the library's default: every IEEE operation of the reference kept (no contraction, correctly "
                     "rounded division and square root) — the oracle's frame bit for bit (tests/test_parity_gpu.py); parity "
                     "forbids contraction, so its attainable ceiling against the FMA-counted peak is 0.5")
        dt, st = timed(r, 5)
        vj, vsrc = replayed_profile("tolerance_pmc_summary.json")
        out["LABELLED_headline_frame_fp64_tolerance"] = row_of(
            dt, st, 5, len(data.object), cfg["width"], d4_cornell,
            pixels_that_differ_from_the_exact_frame=differing, pixels_outside_1e_4=outside, max_abs_delta=float(delta.max()),
            counters_equal_the_exact_kernels=all(st[k] == est[k] for k in ("casts", "bounces", "draws")),
            executed_valu_wave_instructions=({"value": vj["SQ_INSTS_VALU"], "measured_in_run": False, "source": vsrc} if vj else None),
            note="the same kernel source with FMA contraction and division / square root to about one ulp; float islands, RNG, "
                 "roulette thresholds, the unfused fold and the order of the additions kept; primary rays (shared by the S "
                 "samples of a sub-pixel) keep the reference's bits and their exact ties — the Cornell box's wall seams project "
                 "onto the image's diagonals — are settled with the reference's arithmetic.  Asserted <= 1e-4 per pixel against "
                 "the exact frame in tests/test_tolerance_gpu.py; the bit-exact kernel stays the default")

    def fp32_row():
        # SEPARATELY LABELLED row: single precision with the hardware's sqrt / sin / cos (variant 16) — not a parity path;
        # reported with the fraction of pixels that leave the north_star tolerance against the fp64 frame of this run
        data = headline_data()
        exact_frame, _ = rtm.Renderer(data, mode="repaired", max_bounces=cfg["max_bounces"], seed=cfg["seed"], device=device,
                                      host_trig=host_trig).render_rows_device(want=("f32",), stats=True)
        r = rtm.Renderer(data, mode="repaired", max_bounces=cfg["max_bounces"], seed=cfg["seed"], device=device, variant=16)
        fast_frame, _ = r.render_rows_device(want=("f32",), stats=True)
        dt, st = timed(r, 3)
        delta = (fast_frame["f32"].double() - exact_frame["f32"].double()).abs()
        out["LABELLED_headline_frame_fp32_fast_NOT_PARITY"] = row_of(
            dt, st, 3, len(data.object), cfg["width"], d4_cornell, dtype="f32",
            **{"pixels_outside_1e-4_of_the_fp64_frame": float((delta.amax(dim=2) > 1e-4).double().mean())},
            max_abs_delta=float(delta.max()), mean_abs_delta=float(delta.mean()),
            note="float arithmetic, v_sqrt/v_rsq/v_sin/v_cos, fused multiply-adds, forward throughput: a sample whose ray "
                 "grazes a silhouette may take another path than the reference's; not comparable with the headline value")

    def plane_room():
        # scenes/planeRoom.json (png::PlaneObject completed as a finite square) at 1080p x 256 spp: the chunked
        # LDS-table kernel with the plane test in its object chunk
        room = rtm.LoadData(os.path.join(ROOT, "scenes", "planeRoom.json")).data
        room.width, room.height, room.samples, room.superSamples = 1920, 1080, 16, 4
        r = rtm.Renderer(room, mode="repaired", max_bounces=8, seed=cfg["seed"], device=device, host_trig=host_trig)
        dt, st = timed(r, 3)
        out["plane_room_1080p_256spp_max8"] = row_of(
            dt, st, 3, len(room.object), 1920, None,
            note="6 planes + 2 spheres; not a reference scene (the reference's PlaneObject::Intersect is unfinished); the "
                 "roofline prices every object at a sphere's 17 flops per test")

    def c5():
        stress = rtm.make_stress_scene(n=100_000, seed=12345)
        stress.width, stress.height, stress.samples, stress.superSamples = 1920, 1080, 256, 1
        # BASELINE configs[4] as named: the whole 1080p frame at 256 spp, as variant 0 renders it — the uniform-grid
        # kernel: the reference loop's nearest hit for every cast (same image bit for bit, tests/test_grid_gpu.py) from
        # tens of Intersect calls instead of 100 000
        r = rtm.Renderer(stress, mode="repaired", max_bounces=8, seed=cfg["seed"], device=device, host_trig=host_trig)
        dt, st = timed(r, 3)
        # the sphere tests the walks make: one more render of the same frame with the counting instantiation
        _, cst = rtm.Renderer(stress, mode="repaired", max_bounces=8, seed=cfg["seed"], device=device, host_trig=host_trig,
                              count_tests=True).render_rows_device(want=("f32",), stats=True)
        if cst["casts"] == st["casts"]:
            st["object_tests"] = cst["object_tests"]
        row = row_of(dt, st, 3, 100_000, 1920, None, grid_replay(),
                     note="the nearest hit of the reference's loop over all 100 000 spheres (src/Renderer.cpp:58-73) found "
                          "through a uniform grid: the same hit object, distance and image; the rows below are the exhaustive kernel")
        if cpu_rows > 0:
            # the CPU port beside it (timed on a crop and scaled — the reference's loop makes 100 000
            # Intersect calls per cast): a 96x96 block of pixels in the middle of the frame at 4 spp, every host core
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import _oracle
            import numpy as np
            crop = rtm.make_stress_scene(n=100_000, seed=12345)
            crop.width, crop.height, crop.samples, crop.superSamples = 1920, 1080, 4, 1
            cst_, carr, cn = crop.to_c()
            ost = _oracle.Settings.from_buffer_copy(bytes(cst_))
            oarr = (_oracle.Sphere * cn).from_buffer_copy(bytes(carr))
            xy = np.stack(np.meshgrid(np.arange(912, 1008), np.arange(492, 588)), axis=-1).reshape(-1, 2).astype(np.int32)
            threads = _oracle.lib().rtmo_max_threads()
            t0 = time.perf_counter()
            _oracle.render_pixels(ost, oarr, cn, _oracle.make_options(mode=1, max_bounces=8, seed=cfg["seed"], height=1080), xy,
                                  threads=threads)
            cdt = time.perf_counter() - t0
            row["cpu_baseline"] = {"value": len(xy) * 4 / cdt / 1e6, "unit": "Msamples/s", "cores": threads, "kind": "port",
                                   "sample": f"a 96x96 block of the frame's pixels at 4 spp ({len(xy) * 4} samples, {cdt:.2f} s wall, "
                                             "OpenMP per-pixel parallel, the reference's loop over all 100 000 spheres)"}
        out["c5_stress_100k_full_1080p_256spp"] = row

        # the exhaustive large-scene pipeline (variant 12: every cast tests all 100 000 spheres, packed-fp32 rejection
        # in front of the exact test): a strip, and the whole frame (~24 s) — the path north_star names for configs[4]
        rx = rtm.Renderer(stress, mode="repaired", max_bounces=8, seed=cfg["seed"], device=device, host_trig=host_trig,
                          variant=12)
        rx.render_rows_device(508, 516, want=("f32",), stats=True)  # warm: buffers

        def c5_row(lo, hi, replay=None):
            t0 = time.perf_counter()
            o, st = rx.render_rows_device(lo, hi, want=("f32",), stats=True)
            dt = time.perf_counter() - t0
            st["pixels"] = int(o["f32"].shape[0]) * int(o["f32"].shape[1])
            return row_of(dt, st, 1, 100_000, 1920, None, replay)
        row = c5_row(508, 572)
        row["note"] = ("a 64-row strip: 122 880 rays per trip, the sphere list cut into 8 parts so that rays x parts fill the "
                       "chip (DESIGN.md §4)")
        out["c5_stress_100k_exhaustive_rows_508_572_of_1080p_256spp"] = row
        if full_c5:
            replay = None
            tj, tsrc = replayed_profile("c5_traffic.json")
            if tj:
                replay = {"traffic": tj["bytes_per_frame"], "traffic_source": {"measured_in_run": False, "file": tsrc, "note": tj.get("note")}}
            out["c5_stress_100k_exhaustive_full_1080p_256spp"] = c5_row(0, 1080, replay)

    guarded("c2_cornell_512x512_256spp", c2)
    guarded("LABELLED_headline_frame_fp64_tolerance", tolerance_row)
    guarded("LABELLED_headline_frame_with_primary_hit_reuse", reuse_row)
    guarded("LABELLED_headline_frame_fp32_fast_NOT_PARITY", fp32_row)
    guarded("plane_room_1080p_256spp_max8", plane_room)
    guarded("c5_stress_100k", c5)
    return out


def self_launch(n):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: start the N ranks ourselves — BEFORE this
    process imports torch or touches a GPU — through torch.distributed.run (one rank per GPU, rendezvous on
    127.0.0.1 at a free port), pass rank 0's single JSON line through on stdout and return the launcher's status."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("NCCL_SOCKET_IFNAME", "lo")      # N GPUs of ONE node: RCCL's bootstrap over loopback
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # this pool's driver supports dmabuf IPC only
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    os.write(2, f"bench.py: no launcher in the environment, starting {n} ranks: {' '.join(cmd[1:9])} bench.py ...\n".encode())
    return subprocess.run(cmd, env=env).returncode


class stdout_to_stderr:
    """RCCL prints a version banner and gloo a connection note on fd 1 when a group comes up; the contract is ONE JSON
    line on stdout, so fd 1 points at stderr while the group is brought up."""

    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self.saved, 1)
        os.close(self.saved)
        return False


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--width", type=int, default=HEADLINE["width"])
    ap.add_argument("--height", type=int, default=HEADLINE["height"])
    ap.add_argument("--samples", type=int, default=HEADLINE["samples"])
    ap.add_argument("--super-samples", type=int, default=HEADLINE["super_samples"])
    ap.add_argument("--max-bounces", type=int, default=HEADLINE["max_bounces"])
    ap.add_argument("--variant", type=int, default=None,
                    help="kernel variant (rtm_variant_name).  Default: 18 — the fp64 tolerance row: the default kernel's source "
                         "with FMA contraction and one-ulp division / square root, asserted within north_star's 1e-4 per pixel "
                         "(observed: 0 pixels differ on every BASELINE Cornell configuration, tests/test_tolerance_gpu.py) — "
                         "where it serves the workload (c2 / c3 / c4 with a depth cap of at most 8), else 0; the bit-exact "
                         "default kernel's row is measured beside it (other_configs.headline_frame_bit_exact_kernel)")
    ap.add_argument("--exact", action="store_true", help="headline through the bit-exact default kernel (variant 0)")
    ap.add_argument("--workload", default="c3", choices=["c2", "c3", "c4", "c5"],
                    help="BASELINE.json configs: c2 Cornell 512x512x256spp, c3 headline (default), c4 Cornell 4K x 4096spp, "
                         "c5 100k-sphere stress scene 1080p x 256spp (use --rows to bound it)")
    ap.add_argument("--rows", default="", help="render only rows a:b of the frame (value counts those samples)")
    ap.add_argument("--device-trig", action="store_true",
                    help="the device's own sin/cos instead of RTM_MODE_HOST_TRIG (the default: sin/cos exactly as the host "
                         "libm returns them, bit-identical to the oracle on every scene).  ~2 %% faster; the headline frame is "
                         "bit-identical either way (tests/test_parity_gpu.py), adversarial scenes are not")
    ap.add_argument("--host-trig", action="store_true", help="(default; kept for old command lines)")
    ap.add_argument("--no-extras", action="store_true", help="skip with_d2h and other_configs")
    ap.add_argument("--no-full-c5", action="store_true", help="(kept for old command lines: the full frame of the 100k-sphere "
                                                              "scene is a quarter of a second through the grid kernel)")
    ap.add_argument("--full-c5-exhaustive", action="store_true", help="(default since round 4; kept for old command lines)")
    ap.add_argument("--no-full-c5-exhaustive", action="store_true",
                    help="other_configs: skip the full 1080p x 256 spp frame of the 100k-sphere scene through the exhaustive "
                         "pipeline (variant 12, ~24 s: the path north_star names for configs[4]); its 64-row strip is always "
                         "measured")
    ap.add_argument("--layout", default="bands", choices=["bands", "strips"],
                    help="N > 1: interleaved 8-row bands (default) or contiguous row strips per rank")
    ap.add_argument("--ab", type=str, default="", help="comma-separated variants: interleaved A/B rounds, kernel ms each")
    ap.add_argument("--backend", default=None, choices=["nccl", "gloo"],
                    help="process-group backend (default nccl = RCCL when N > 1); gloo + --same-device rehearses the "
                         "N-rank path on one GPU; given explicitly with N = 1 the one-rank group is created too and "
                         "every step goes through the gather")
    ap.add_argument("--same-device", action="store_true", help="every rank uses cuda:0 (rehearsal only)")
    ap.add_argument("--cpu-rows", type=int, default=64, help="rows of the CPU baseline sample (0 = skip)")
    ap.add_argument("--stage-timeout", type=float, default=300.0,
                    help="seconds a start-up / collective stage may take before the rank exits with code 3 naming it "
                         "(render stages get 10x this)")
    ap.add_argument("--launch-check", action="store_true",
                    help="bring the N ranks and their process group up, all-reduce the ranks, print one JSON line and stop "
                         "(no GPU touched: the CPU rehearsal of the launch path)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus))  # nothing below has run yet: no torch import, no GPU call in this process

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    from raytracingmin_amd.distributed import StageWatchdog
    dog = StageWatchdog(limit_s=args.stage_timeout, rank=rank, quiet=(world == 1 and args.backend is None))
    dog.enter("import torch", max(600.0, args.stage_timeout))  # a fresh box pages the image in: minutes, with 8 ranks at once
    import torch
    import torch.distributed as dist
    import raytracingmin_amd as rtm

    args.gpus = world  # the launcher's world size is what runs
    if args.launch_check:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        with stdout_to_stderr():
            dog.enter(f"init_process_group({args.backend or 'gloo'}, world {world})")
            dist.init_process_group(args.backend or "gloo", rank=rank, world_size=world)
            dog.enter("all_reduce of the ranks")
            t = torch.zeros(world, dtype=torch.int64)
            t[rank] = rank + 1
            dist.all_reduce(t)
            dog.enter("destroy_process_group")
            dist.destroy_process_group()
        dog.done()
        if rank == 0:
            print(json.dumps({"launch_check": True, "n_gpus": world, "ranks_seen": [int(v) - 1 for v in t],
                              "backend": args.backend or "gloo"}), flush=True)
        return
    if args.same_device:
        local_rank = 0
    dog.enter(f"torch.cuda.set_device({local_rank})")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_group = world > 1 or args.backend is not None
    backend = args.backend or "nccl"
    if use_group:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")  # N GPUs of ONE node: RCCL's bootstrap over loopback
        with stdout_to_stderr():  # until the group has done its first collective
            dog.enter(f"init_process_group({backend}, world {world}, NCCL_SOCKET_IFNAME={os.environ['NCCL_SOCKET_IFNAME']})")
            if backend == "nccl":
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
            else:
                dist.init_process_group("gloo", rank=rank, world_size=world)
            dog.enter("first barrier (communicator bring-up)")
            dist.barrier()
            if backend == "nccl":
                torch.cuda.synchronize(dev)
    host_trig = not args.device_trig
    dog.enter("scene load + renderer set-up")
    if args.variant is None:
        tol_serves = args.workload in ("c2", "c3", "c4") and args.samples * args.super_samples ** 2 < 65536 and not args.exact
        args.variant = 18 if tol_serves else 0

    cfg = dict(HEADLINE, width=args.width, height=args.height, samples=args.samples,
               super_samples=args.super_samples, max_bounces=args.max_bounces)
    if args.workload == "c2":
        cfg.update(width=512, height=512, samples=16, super_samples=4)
    elif args.workload == "c4":
        cfg.update(width=3840, height=2160, samples=256, super_samples=4)
    elif args.workload == "c5":
        cfg.update(scene="stress-100k (SURVEY App. D, seed 12345)", width=1920, height=1080, samples=256, super_samples=1)
    if args.workload == "c5":
        data = rtm.make_stress_scene(n=100_000, seed=12345)
    else:
        scene = os.path.join(ROOT, "scenes", cfg["scene"])
        data = rtm.LoadData(scene).data
    data.width, data.height = cfg["width"], cfg["height"]
    data.samples, data.superSamples = cfg["samples"], cfg["super_samples"]
    spp = cfg["samples"] * cfg["super_samples"] ** 2
    row_lo, row_hi = 0, cfg["height"]
    if args.rows:
        row_lo, row_hi = (int(v) for v in args.rows.split(":"))

    from raytracingmin_amd.distributed import StripRenderer
    sr = StripRenderer(data, rank=rank, world=world, device=local_rank, mode=cfg["mode"],
                       max_bounces=cfg["max_bounces"], seed=cfg["seed"], variant=args.variant,
                       rows=(row_lo, row_hi), layout=args.layout, host_trig=host_trig,
                       force_collective=use_group and world == 1)

    if args.ab:
        # interleaved rounds in ONE process (guide rule 24): median/min kernel ms per variant
        from raytracingmin_amd.renderer import Renderer
        vs = [int(v) for v in args.ab.split(",")]
        rs = {v: Renderer(data, mode=cfg["mode"], max_bounces=cfg["max_bounces"], seed=cfg["seed"],
                          device=local_rank, variant=v, host_trig=host_trig) for v in vs}
        times = {v: [] for v in vs}
        for rnd in range(args.steps + args.warmup):
            for v in vs:
                _, st = rs[v].render_rows_device(want=("f32",), stats=True)
                if rnd >= args.warmup:
                    times[v].append(st["kernel_ms"])
        for v in vs:
            t = sorted(times[v])
            print(json.dumps({"variant": v, "name": rtm.lib().rtm_variant_name(v).decode(),
                              "kernel_ms_median": t[len(t) // 2], "kernel_ms_min": t[0],
                              "Msamples_per_s_median": cfg["width"] * cfg["height"] * spp / t[len(t) // 2] / 1e3}), flush=True)
        return

    def barrier():
        if use_group:
            dist.barrier()
        torch.cuda.synchronize(dev)

    dog.enter(f"{args.warmup} warm-up step(s) + one instrumented step (render + gather)", 10 * args.stage_timeout)
    for _ in range(args.warmup):
        sr.step()
    # kernel time and counters of one instrumented launch (outside the timed region)
    barrier()
    stats = sr.step(stats=True)
    barrier()

    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
          for _ in range(args.steps)]
    barrier()
    dog.enter(f"{args.steps} timed step(s) (render + gather)", 10 * args.stage_timeout)
    t0 = time.perf_counter()
    for k in range(args.steps):
        sr.step(events=ev[k])
    barrier()
    elapsed = time.perf_counter() - t0

    dog.enter("all_reduce(MAX) of the ranks' times")
    kernel_ms = sum(a.elapsed_time(b) for a, b in ev) / max(1, args.steps)
    kernel_ms_local = kernel_ms
    t = torch.tensor([elapsed, kernel_ms], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
    if use_group:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed, kernel_ms = float(t[0]), float(t[1])
    sr.renderer.stream_status()  # the timed steps ran without rtm_stats: a truncated path would surface here

    # SURVEY.md §8(d) wall time: kernel(s) + the final D2H of the frame (rank 0 holds it after the gather)
    with_d2h = None
    dog.enter("extras: with_d2h steps, CPU baseline, other configs", 30 * args.stage_timeout)
    if not args.no_extras:
        host = torch.empty(sr.image.shape, dtype=sr.image.dtype, pin_memory=True) if rank == 0 and sr.image is not None else None
        barrier()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            sr.step()
            if host is not None:
                host.copy_(sr.image, non_blocking=True)
        barrier()
        with_d2h = time.perf_counter() - t1

    # an N-rank line proves itself, outside the timed region: the assembled frame against rank 0's own
    # single-launch render bit for bit, every rank's kernel time / rows / counters, the gather alone timed
    evidence = None
    if use_group:
        dog.enter("multi-GPU evidence: per-rank stats, the gather alone, rank 0's single-GPU frame", 30 * args.stage_timeout)
        from raytracingmin_amd.distributed import multi_gpu_evidence
        evidence = multi_gpu_evidence(sr, stats, kernel_ms_local, args.steps, barrier, use_group=True)

    total_samples = cfg["width"] * (row_hi - row_lo) * spp
    value = total_samples * args.steps / elapsed / 1e6

    if rank == 0:
        n_spheres = len(data.object)
        cps = stats["casts"] / stats["samples"]
        bps = stats["bounces"] / stats["samples"]
        cpu = None
        extras_failed = {}
        d4 = 0.93
        if world == 1 and args.cpu_rows > 0:
            try:
                cpu = cpu_baseline(cfg, args.cpu_rows)
                d4 = cpu["d4_fraction"]
            except Exception as exc:  # the headline line must come out whatever an extra does; the failure is named in it
                cpu, extras_failed = None, {"cpu_baseline": repr(exc)}
        extras_failed = dict(extras_failed)
        pk = None
        if world == 1 and not args.no_extras:
            try:
                pk = measured_fp64_peak(rtm)
            except Exception as exc:
                extras_failed["peak_measured"] = repr(exc)
        rows0 = stats["samples"] // (cfg["width"] * spp)  # rows this rank's launch stores
        stats["pixels"] = rows0 * cfg["width"]
        traffic, traffic_src = None, None
        tpath = os.path.join(ROOT, "profiles", "latest_traffic.json")
        headline = args.workload == "c3" and not args.rows and \
            all(cfg[k] == HEADLINE[k] for k in ("width", "height", "samples", "super_samples", "max_bounces"))
        replay = None
        if args.variant == 18:
            tpath = os.path.join(ROOT, "profiles", "r4", "tolerance_traffic.json")
        if world == 1 and headline and args.variant in (0, 18) and os.path.exists(tpath):
            tj = json.load(open(tpath))
            replay = {"traffic": tj["bytes_per_launch"],  # PMC bytes of a committed rocprofv3 --pmc pass of this command
                      "traffic_source": {"measured_in_run": False, "file": os.path.relpath(tpath, ROOT),
                                         "profile": tj.get("source", "profiles/r1/default_pmc_summary.json"),
                                         "what": "FETCH_SIZE + WRITE_SIZE per launch, corrected as MI355X_MICROARCH.md prescribes",
                                         "note": tj.get("note")}}
        if stats["variant"] == 17:
            replay = grid_replay()
            if world == 1 and not args.no_extras:  # the walks' sphere tests: one counting render of this rank's rows
                try:
                    from raytracingmin_amd.renderer import Renderer
                    _, cst = Renderer(data, mode=cfg["mode"], max_bounces=cfg["max_bounces"], seed=cfg["seed"], device=local_rank,
                                      host_trig=host_trig, count_tests=True).render_rows_device(row_lo, row_hi, want=("f32",), stats=True)
                    if cst["casts"] == stats["casts"]:
                        stats["object_tests"] = cst["object_tests"]
                except Exception as exc:
                    extras_failed["object_tests"] = repr(exc)
        resolved = rtm.lib().rtm_variant_name(stats["variant"]).decode()  # what the library ran (rtm_stats.variant)
        # the Cornell scenes' D4 >= 0 share comes from this run's oracle sample; other scenes' is not measured
        d4_here = None if args.workload == "c5" else d4
        roof = roofline_of(stats, kernel_ms, n_spheres, cfg["width"], d4_here, pk["value"] if pk else None, replay)
        roof["achieved_source"] = {"measured_in_run": True, "what": "the work model of the kernel the library reports "
                                   "(rtm_stats.variant; roofline_of in bench.py) x this run's kernel counters / HIP-event kernel time"}
        if roof["bound"] == "valu-fp64":
            roof["note"] = ("no MFMA and not HBM-bound: ~1 kflop fp64 per sample vs 0.012 B of HBM traffic; peak counts FMA as 2 "
                            "flops" + ("; this is the labelled tolerance row, compiled with contraction" if stats["variant"] == 18 else
                                       " but parity forbids contraction, so the attainable ceiling is <= 0.5"))
        line = {
            "metric": "Msamples/s (W*H*spp/s), Cornell box 1080p@1024spp" if args.workload == "c3" else
                      f"Msamples/s (W*H*spp/s), BASELINE configs workload {args.workload}", "value": value,
            "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{cfg['scene']} rows {row_lo}:{row_hi} of {cfg['width']}x{cfg['height']} "
                                   f"{spp}spp (SS {cfg['super_samples']} x S {cfg['samples']}), L1 repaired, "
                                   f"max_bounces {cfg['max_bounces']}, seed 0x5EED, "
                                   f"{'host-libm sin/cos (RTM_MODE_HOST_TRIG)' if host_trig else 'device sin/cos (--device-trig)'}, "
                                   f"{'interleaved 8-row bands' if args.layout == 'bands' else 'row strips'} over "
                                   f"{world} GPU(s) + one gather",
                       "variant": resolved, "sample_split_waves_per_tile": stats["split"],
                       "collective": (f"{backend} gather, {world} rank(s)" if use_group else "none (one rank, frame stays in HBM)"),
                       "casts_per_sample": cps, "bounces_per_sample": bps},
            "roofline": roof,
        }
        if evidence is not None:
            line["frame_matches_single_gpu"] = evidence["frame_matches_single_gpu"]
            line["gather_ms"] = evidence["gather_ms"]
            line["config"]["per_rank"] = evidence["per_rank"]
            line["config"]["multi_gpu_evidence"] = {k: evidence[k] for k in ("totals", "kernel_ms_slowest_over_mean", "how")}
        ppath = next((q for q in (os.path.join(ROOT, "profiles", "r4", "default_pmc_summary.json"),
                                  os.path.join(ROOT, "profiles", "r3", "default_pmc_summary.json"),
                                  os.path.join(ROOT, "profiles", "r2", "default_pmc_summary.json"),
                                  os.path.join(ROOT, "profiles", "r1", "default_pmc_summary.json")) if os.path.exists(q)), "")
        if args.variant == 18:
            ppath = os.path.join(ROOT, "profiles", "r4", "tolerance_pmc_summary.json")
            ppath = ppath if os.path.exists(ppath) else ""
        if world == 1 and headline and args.variant in (0, 18) and ppath:
            line["roofline"]["valu_issue"] = valu_issue_view(json.load(open(ppath)), os.path.relpath(ppath, ROOT))
        if pk is not None:
            line["roofline"]["peak_measured"] = pk
            line["roofline"]["peak_source"] = "AMD datasheet, FP64 vector 78.6 TFLOP/s (MI355X_MICROARCH.md lists fp32 157.3 only)"
        if cpu is not None:
            # the oracle is a restatement ("port"); what it is worth against the genuine compiled reference was
            # measured once in the build container (the GPU box never sees /root/reference): DESIGN.md §8
            cpu["oracle_vs_reference_ratio"] = {
                "value": 0.89, "measured_in_run": False,
                "provenance": "round-1 build container, Xeon 2.1 GHz, 1 thread, Cornell 256x256x64spp uncapped: oracle 1.01 vs "
                              "the compiled reference 1.125 Msamples/s (SURVEY.md §6); L0: 4.03 vs 3.44 (ratio 1.17)"}
            line["cpu_baseline"] = cpu
        if with_d2h is not None:
            line["with_d2h"] = {"value": total_samples * args.steps / with_d2h / 1e6, "unit": "Msamples/s",
                                "ms_per_step": with_d2h / args.steps * 1e3, "measured_in_run": True,
                                "what": "the same K steps with the gathered frame copied to pinned host memory each step "
                                        "(SURVEY.md §8d wall time = kernel + final D2H/gather); not the headline value"}
        if world == 1 and not args.no_extras and headline:
            try:
                line["other_configs"] = other_configs(rtm, cfg, local_rank, host_trig, full_c5=not args.no_full_c5_exhaustive,
                                                      cpu_rows=args.cpu_rows, d4_cornell=(cpu["d4_fraction"] if cpu else None),
                                                      peak_measured=(pk["value"] if pk else None), headline_variant=args.variant)
            except Exception as exc:
                extras_failed["other_configs"] = repr(exc)
        if extras_failed:
            line["extras_failed"] = extras_failed
        print(json.dumps(line), flush=True)
    if use_group:
        dog.enter("destroy_process_group")
        dist.destroy_process_group()
    dog.done()


if __name__ == "__main__":
    main()
