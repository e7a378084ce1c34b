#!/usr/bin/env python3
"""bench.py -- throughput of the SIFT extraction path on MI355X.

Metric: 1080p images/s (keypoints + descriptors), whole job over N GPUs, plus the HBM
rate of the blur kernels.  Workload: 1920x1080 synthetic frames (5-octave value noise, seeds 1000...),
resident in HBM before the timed region; one step = one batch of `--batch` frames through the whole path
(gray/upsample/blur, Gaussian pyramid, fused DoG+extrema, refinement, orientation, descriptors, scale
ordering, 144-byte packed features).  With the defaults 4 steps x 256 frames (the 256-frame batch configuration).

  python bench.py --gpus 1 --steps 4 --warmup 2
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

N > 1: one process per GPU, frames sharded by rank as in the 4096-frame config (global frame f has seed 1000 + f and
lives on rank f // 512; no data-path exchange during extraction); after every step the ranks all-gather their packed
feature sets (sg_allgather_features of the C ABI: an all-gather-v over RCCL, overlapped with the next step's
kernels).  Before a multi-rank measurement sets anything up, a trial of that exchange runs in child processes beside
the ranks (preflight_abi_exchange); if it fails or does not return, the measured run exchanges through
torch.distributed instead and says so (exchange, exchange_preflight).  Rank 0 prints one JSON line.  Outside the timed region the line also gets: a check of the last batch
against the same frames run alone, single-frame latencies (1080p and 8K), and the throughput of the reference's own
entry point Sift() through the host C++ layer.
"""
import argparse
import json
import os
import re
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E peak
OCTAVE_PX_1080P = None


def octave_pixels(w, h):
    """Sum over octaves of pixels per level, base = 2w x 2h (reference sift.cpp:1653 octave rule)."""
    rows, cols = 2 * h, 2 * w
    octaves = int(np.log(float(min(rows, cols))) / np.log(2.0) - 2)
    px = []
    for o in range(octaves):
        px.append(rows * cols)
        if rows > 1 and cols > 1:
            rows, cols = rows // 2, cols // 2
    return px


def octave_dims(w, h):
    """(rows, cols) of every octave, base = 2w x 2h."""
    rows, cols = 2 * h, 2 * w
    octaves = int(np.log(float(min(rows, cols))) / np.log(2.0) - 2)
    out = []
    for o in range(octaves):
        out.append((rows, cols))
        if rows > 1 and cols > 1:
            rows, cols = rows // 2, cols // 2
    return out


def measured_stream_ceilings():
    """What plain streams reach on this part, from the committed outputs of the micro-benchmarks under
    profiles/ubench/ (GB/s): read-only, write-only and one-read-one-write rates of persistent loops
    (stream_ceilings.hip), the copy whose workgroups take one 4 KB piece each and end -- the shape that reaches the
    part's 6.3 TB/s -- and the best persistent copy (copy_ceiling.hip), and the strip walk of the blur kernels at
    segment lengths a column filter can use (stream_ceilings.hip with short segments).  The yardsticks the pyramid
    kernels' rates are read against beside the nominal 8 TB/s."""
    import glob
    out = {}
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "stream_ceilings.txt")), reverse=True):
        try:
            for line in open(path):
                m = re.match(r"(linear copy|read only|write only)\s+[\d.]+ ms\s+([\d.]+) TB/s", line)
                if m:
                    out[m.group(1).replace(" ", "_")] = float(m.group(2)) * 1e3
        except OSError:
            continue
        if out:
            out["source"] = os.path.relpath(path, ROOT)
            break
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "copy_ceiling_8g.txt")), reverse=True):
        try:
            one_trip, persistent = 0.0, 0.0
            for line in open(path):
                m = re.match(r"(.+?)\s+avg [\d.]+ ms ([\d.]+) TB/s", line)
                if not m:
                    continue
                rate = float(m.group(2)) * 1e3
                if m.group(1).startswith("one trip per thread U1 "):
                    one_trip = rate
                elif m.group(1).startswith(("grid-stride", "chunk")):
                    persistent = max(persistent, rate)
            if one_trip:
                out["copy_one_4KB_piece_per_workgroup"] = one_trip
                out["copy_persistent_loops_best"] = persistent
                out["copy_source"] = os.path.relpath(path, ROOT)
                break
        except OSError:
            continue
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "strip_short.txt")), reverse=True):
        try:
            best = 0.0
            for line in open(path):
                m = re.match(r"NW\d+ V\d+ PF\d+ nt\d remap0 seg (\d+)\s+[\d.]+ ms\s+([\d.]+) TB/s", line)
                if m and int(m.group(1)) >= 32:
                    best = max(best, float(m.group(2)) * 1e3)
            if best:
                out["strip_walk_segments_of_32_rows_and_more_best"] = best
                out["strip_source"] = os.path.relpath(path, ROOT)
                break
        except OSError:
            continue
    return out or None


def measured_traffic(kernel, tname, batch, width, height):
    """HBM bytes per launch of `kernel` from the committed PMC passes (profiles/r*/traffic.json, written by
    profiles/summarise.py from separate FETCH_SIZE and WRITE_SIZE runs of this same command); None when no pass
    was collected for this kernel at this workload."""
    import glob
    m = re.match(r"(blur_\w+_k)<(\w+),R=(\d+)(,upsampling)?>", kernel)
    d = re.match(r"detect_k<(\w+)>", kernel)
    if not m and not d:
        return None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "traffic.json")), reverse=True):
        try:
            t = json.load(open(path))
        except (OSError, ValueError):
            continue
        cfg = t.get("_config", {})
        if (cfg.get("batch"), cfg.get("width"), cfg.get("height")) != (batch, width, height):
            continue
        for name, rec in t.items():
            if d and name.startswith("sg::detect_k<%s," % d.group(1)):
                return rec["hbm_bytes_per_launch"]
            if m and name.startswith("sg::%s<%s, %s," % (m.group(1), m.group(2), m.group(3))):
                targs = [a.strip() for a in name[name.index("<") + 1: name.rindex(">")].split(",")]
                if targs[-1] == "true":  # the fused multiply-add instantiation: another mode's launches
                    continue
                # blur_win_k<T, R, V, UPS, GDMA, F>: the fourth argument marks the upsampling form
                if m.group(1) != "blur_win_k" or (len(targs) > 3 and targs[3] == "true") == bool(m.group(4)):
                    return rec["hbm_bytes_per_launch"]
    return None


# ns of one SIMD per wave-instruction under full load, measured with profiles/ubench/valu_rate.hip on the MI355X boxes
# used here (profiles/r02/valu_rate.txt): double-precision arithmetic, conversions and compares 2.1; v_rsq_f64 / v_rcp_f64
# 7.4; the other vector instructions between 1.1 (VOP1/VOP2 moves and logic) and 1.9 (three-operand integer forms,
# selects): priced at 1.9.  LDS: ns of a CU's LDS pipe per 64-lane instruction: ds_add_u64 3.5, ds_add_f64 6.8
# (distinct addresses; same-address lanes serialise), reads 1.8.
ISSUE_NS = {"valu_f64": 2.1, "transcendental_f64": 7.4, "valu_other": 1.9}
LDS_NS = {"lds_u64_adds": 3.5, "lds_f64_adds": 6.8, "lds_other": 1.8}
# What an LDS instruction costs the SIMD that issues it, beside double-precision work of the other waves
# (profiles/ubench/issue_mix.hip, profiles/r03/issue_mix.txt: 32 ds_read_b64 between 32 v_fma_f64 add 2.4 ns each; the
# 64-bit adds from the descriptor kernel run with and without them, profiles/r03/descr_variants.txt: 5-6.7 ms over
# 150.8 M trips of 8 adds on 1024 SIMDs = 4.2-5.7 ns each).
SIMD_NS_LDS = {"adds": 4.2, "other": 2.4}


def valu_roofline(kernel, trips, kernel_ms, n_simd=1024, n_cu=256):
    """Prices a histogram kernel against the vector issue rate and the LDS pipe: static instruction counts of its
    sample loop (profiles/isa_counts.json, from the ISA) x measured 64-sample trips x the measured cost of an
    instruction, against the kernel's measured time.  The loop's per-feature set-up is not counted, so the fractions
    are lower bounds.  simd_issue_* adds what the loop's LDS instructions cost the issuing SIMD: the kernel's time
    follows the sum of what a SIMD has to issue, not the busiest pipe."""
    try:
        isa = json.load(open(os.path.join(ROOT, "profiles", "isa_counts.json")))[kernel]
    except (OSError, ValueError, KeyError):
        return None
    if not trips or kernel_ms <= 0:
        return None
    f64 = isa["valu_f64"] - isa["transcendental_f64"]
    valu_ns = f64 * ISSUE_NS["valu_f64"] + isa["transcendental_f64"] * ISSUE_NS["transcendental_f64"] + \
        isa["valu_other"] * ISSUE_NS["valu_other"]
    n_adds = isa.get("lds_u64_adds", 0) + isa.get("lds_f64_adds", 0)
    adds = isa.get("lds_u64_adds", 0) * LDS_NS["lds_u64_adds"] + isa.get("lds_f64_adds", 0) * LDS_NS["lds_f64_adds"]
    lds_other = isa["lds"] - n_adds
    lds_ns = adds + lds_other * LDS_NS["lds_other"]
    simd_ns = valu_ns + n_adds * SIMD_NS_LDS["adds"] + lds_other * SIMD_NS_LDS["other"]
    issue_ms = trips * valu_ns / n_simd * 1e-6
    lds_ms = trips * lds_ns / n_cu * 1e-6
    simd_ms = trips * simd_ns / n_simd * 1e-6
    return {"kernel": kernel, "trips_64_samples": int(trips), "kernel_ms": kernel_ms,
            "valu_instructions_per_trip": isa["valu_f64"] + isa["valu_other"], "valu_f64_per_trip": isa["valu_f64"],
            "valu_issue_ms": issue_ms, "valu_frac": issue_ms / kernel_ms,
            "lds_instructions_per_trip": isa["lds"], "lds_pipe_ms": lds_ms, "lds_frac": lds_ms / kernel_ms,
            "simd_issue_ms": simd_ms, "simd_issue_frac": simd_ms / kernel_ms}


def batch_check_and_latency(sift_amd, torch, dev, dev_index, prec, W, H, cap, frames, frame_bytes, first, B, out, counts):
    """Outside the timed region: every frame of the last timed batch run alone (a context of one frame) must give the
    same count and the same packed bytes as it did inside the batch (Sift() is a function of one frame's pixels,
    reference sift/sift.cpp:1646-1707).  The same context then times single frames: the 1080p frame of the
    single-image config and, in a context of its own, the 8K one."""
    one = sift_amd.SiftGPU(max_width=W, max_height=H, max_batch=1, precision=prec, device=dev_index)
    tmp = torch.empty(cap * 144, dtype=torch.uint8, device=dev)
    cnt1 = torch.zeros(1, dtype=torch.int32, device=dev)
    bad_counts = bad_bytes = 0
    total1 = 0
    for i in range(B):
        one.extract_batch_device(frames.data_ptr() + (first + i) * frame_bytes, 1, W, H, 3, tmp.data_ptr(), cap,
                                 cnt1.data_ptr())
        one.sync()
        n = int(cnt1.item())
        total1 += n
        if n != int(counts[i]):
            bad_counts += 1
        elif not torch.equal(tmp[: n * 144], out[i * cap * 144: i * cap * 144 + n * 144]):
            bad_bytes += 1
    verify = {"frames": B, "features_batch": int(counts.astype(np.int64).sum()), "features_frames_alone": total1,
              "frames_with_other_count": bad_counts, "frames_with_other_bytes": bad_bytes,
              "ok": bad_counts == 0 and bad_bytes == 0}

    def median_ms(fn, reps):
        t = []
        for _ in range(reps):
            t0 = time.perf_counter()
            fn()
            t.append(1e3 * (time.perf_counter() - t0))
        return float(np.median(t))

    def one_frame(ctx, w, h, d_img, d_out, c, d_cnt):
        ctx.extract_batch_device(d_img, 1, w, h, 3, d_out, c, d_cnt)
        ctx.sync()

    latency = {}
    if (W, H) == (1920, 1080):
        args1 = (one, W, H, frames.data_ptr() + first * frame_bytes, tmp.data_ptr(), cap, cnt1.data_ptr())
        one_frame(*args1)
        latency["1080p_frame_alone_device"] = median_ms(lambda: one_frame(*args1), 20)
    one.close()
    del tmp
    try:  # the 8K single image (7680x4320, 11 octaves, 15360x8640 base)
        w8, h8, cap8 = 7680, 4320, 400000
        big = sift_amd.SiftGPU(max_width=w8, max_height=h8, max_batch=1, precision=prec, device=dev_index)
        img8 = torch.empty(w8 * h8 * 3, dtype=torch.uint8, device=dev)
        out8 = torch.empty(cap8 * 144, dtype=torch.uint8, device=dev)
        big.synth_vnoise5(img8.data_ptr(), w8, h8, 1234)
        args8 = (big, w8, h8, img8.data_ptr(), out8.data_ptr(), cap8, cnt1.data_ptr())
        one_frame(*args8)
        latency["8k_frame_alone_device"] = median_ms(lambda: one_frame(*args8), 5)
        latency["8k_features"] = int(cnt1.item())
        big.close()
    except Exception as e:  # a report, never a dependency of the measured path
        latency["8k_error"] = repr(e)
    return verify, latency


def host_api_throughput(W, H, dev_index, cpu_threads):
    """Sift(const Mat &, vector<Keypoint> &) (reference sift/sift.h:139) through the host C++ layer, as a compiled
    program: latency of one call, and calls per second with one context per host thread."""
    import subprocess
    exe = os.path.join(ROOT, "sift_amd", "host", "sift_host_bench")
    if not os.path.exists(exe):
        return {"error": "%s is not built" % exe}
    threads = cpu_threads or min(16, os.cpu_count() or 1)

    def run(shared):
        p = subprocess.run([exe, str(W), str(H), str(threads), "20", str(dev_index), str(shared)], capture_output=True,
                           text=True, timeout=240)
        if p.returncode != 0:
            raise RuntimeError(p.stderr.strip()[-300:])
        return json.loads(p.stdout.strip().splitlines()[-1])

    try:
        r = run(0)
        r["what"] = ("Sift() on host images, 1120-byte records appended to a host vector; sift_ms_single = one thread, "
                     "one call at a time; images_per_s = %d host threads, one context each, one GPU; "
                     "images_per_s_shared_context = the same threads through the shared batching contexts (two per "
                     "device, up to 16 images per batch: SiftSetBatching(16))" % threads)
        r["images_per_s_shared_context"] = run(16)["images_per_s"]
        return r
    except Exception as e:
        return {"error": repr(e)}


def packed_rows(buf, n):
    """n packed 144-byte features as a sorted array of byte strings (a feature set, whatever its order)."""
    a = buf[: n * 144].cpu().numpy().reshape(n, 144)
    return np.sort(np.ascontiguousarray(a).view([("b", "V144")]).ravel())


def multiset_common(a, b):
    """Number of elements two arrays have in common, counted with multiplicity (the path keeps duplicate keypoints,
    as the reference does, so a frame's feature list may hold the same record more than once)."""
    ua, ca = np.unique(a, return_counts=True)
    ub, cb = np.unique(b, return_counts=True)
    _, ia, ib = np.intersect1d(ua, ub, assume_unique=True, return_indices=True)
    return int(np.minimum(ca[ia], cb[ib]).sum())


def fma_mode_agreement(sift_amd, torch, dev, dev_index, W, H, cap, frames, frame_bytes, first, B, out, counts):
    """The first frames of the last timed batch again with fused multiply-adds in the Gaussian filters (SG_F64_FMA),
    compared with what the timed run (the reference's separate multiplies and adds) produced for them: frames whose
    feature count is the same, features identical byte for byte (position, scale and orientation as packed floats,
    all 128 descriptor bins), and features identical in everything but the packed orientation (an orientation that
    differs by 1e-13 rad now and then rounds to the neighbouring float)."""
    n = min(16, B)
    ctx = sift_amd.SiftGPU(max_width=W, max_height=H, max_batch=n, precision=sift_amd.SG_F64_FMA, device=dev_index)
    out2 = torch.empty(n * cap * 144, dtype=torch.uint8, device=dev)
    cnt2 = torch.zeros(n, dtype=torch.int32, device=dev)
    ctx.extract_batch_device(frames.data_ptr() + first * frame_bytes, n, W, H, 3, out2.data_ptr(), cap, cnt2.data_ptr())
    ctx.sync()
    c2 = cnt2.cpu().numpy()
    same_count = n_exact = n_fma = identical = same_keys = same_but_ori = 0
    for i in range(n):
        a = packed_rows(out[i * cap * 144:], int(counts[i]))
        b = packed_rows(out2[i * cap * 144:], int(c2[i]))
        same_count += int(len(a) == len(b))
        n_exact += len(a)
        n_fma += len(b)
        identical += multiset_common(a, b)
        ra = np.frombuffer(a.tobytes(), np.uint8).reshape(-1, 144)
        rb = np.frombuffer(b.tobytes(), np.uint8).reshape(-1, 144)
        # position and scale (the keypoint), and everything but the packed orientation (bytes 12..15)
        same_keys += multiset_common(ra[:, :12].copy().view([("k", "V12")]).ravel(), rb[:, :12].copy().view([("k", "V12")]).ravel())
        wa = np.concatenate([ra[:, :12], ra[:, 16:]], axis=1).copy().view([("k", "V140")]).ravel()
        wb = np.concatenate([rb[:, :12], rb[:, 16:]], axis=1).copy().view([("k", "V140")]).ravel()
        same_but_ori += multiset_common(wa, wb)
    ctx.close()
    return {"frames": n, "frames_with_same_count": same_count, "features_exact": n_exact, "features_fma": n_fma,
            "features_identical_bytes": identical,
            "features_identical_but_for_the_packed_orientation": same_but_ori,
            "keypoints_identical": same_keys == n_exact == n_fma,      # position and scale of every feature
            "bins_identical": same_but_ori == n_exact == n_fma,        # ... and all 128 descriptor bins
            "orientations_differing_as_packed_floats": same_but_ori - identical}


def fma_mode_throughput(args, B):
    """The default workload once more, in a run of its own, with SG_F64_FMA: frames per second and the blur kernels'
    rates when the filters are no longer bound by separate double-precision multiplies and adds."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--batch", str(B), "--steps", str(args.steps),
           "--warmup", str(args.warmup), "--width", str(args.width), "--height", str(args.height),
           "--precision", "f64_fma", "--cap", str(args.cap), "--no-extras", "--no-cpu-baseline"]
    try:
        p = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
        if p.returncode != 0:
            return {"error": p.stderr.strip()[-300:]}
        r = json.loads(p.stdout.strip().splitlines()[-1])
        return {"images_per_s": r["value"], "ms_per_step": r["ms_per_step"],
                "all_blur_launches_GBps": r["roofline"]["all_blur_launches_GBps"],
                "all_blur_GBps_level_by_level_bytes": r["roofline"]["all_blur_GBps_level_by_level_bytes"],
                "blur_kernels_GBps": {k: v["GBps"] for k, v in r["blur_kernels"].items()},
                "blur_ms_per_step": {k: v for k, v in r["stage_ms_per_step"].items() if k.startswith(("blur", "pyr"))}}
    except Exception as e:
        return {"error": repr(e)}


def host_multi_gpu_driver(W, H, n_gpus, frames):
    """SiftBatchMultiGpu (host C++, sift_amd/host/sift_multi.cpp): host frames sharded over the GPUs by one thread per
    GPU, feature sets all-gathered over RCCL, results back on the host -- as a compiled program."""
    import subprocess
    exe = os.path.join(ROOT, "sift_amd", "host", "sift_multi_check")
    if not os.path.exists(exe):
        return {"error": "%s is not built" % exe}
    try:
        p = subprocess.run([exe, "bench", str(W), str(H), str(frames), str(n_gpus), "32"], capture_output=True, text=True,
                           timeout=240)
        if p.returncode != 0:
            return {"error": (p.stderr.strip() or p.stdout.strip())[-300:]}
        return json.loads(p.stdout.strip().splitlines()[-1])
    except Exception as e:
        return {"error": repr(e)}


def two_batches_in_flight(args, B):
    """The same workload with two contexts of half the batch each, so that the histogram kernels of one batch
    (vector-ALU work) run beside the pyramid kernels of the other (memory work).  Kernel durations measured that way no
    longer describe a kernel running alone, which is why `value` and `roofline` come from the one-context run; this
    is a second, separate run of this script and is reported beside them."""
    import subprocess
    if B < 2:
        return None
    cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--contexts", "2", "--batch", str(B // 2),
           "--steps", str(2 * args.steps), "--warmup", str(2 * args.warmup), "--width", str(args.width),
           "--height", str(args.height), "--precision", args.precision, "--cap", str(args.cap),
           "--no-extras", "--no-cpu-baseline", "--no-profile"]
    try:
        p = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
        if p.returncode != 0:
            return {"error": p.stderr.strip()[-300:]}
        r = json.loads(p.stdout.strip().splitlines()[-1])
        return {"images_per_s": r["value"], "contexts": 2, "batch_per_context": r["config"]["batch"],
                "ms_per_step": r["ms_per_step"], "steps": r["steps"]}
    except Exception as e:
        return {"error": repr(e)}


def preflight_child():
    """One rank of the trial exchange, in a process of its own (`bench.py --preflight-child`).

    With several ranks the C ABI's exchange (sg_comm_create, sg_allgather_features: the library's own RCCL binding)
    meets real peers for the first time, and a collective that never returns cannot be left from inside the process
    that entered it.  So before the measured processes allocate anything, every rank starts this program: the ranks
    meet over gloo on a port of their own, build an RCCL communicator through the C ABI and run the all-gather-v in
    both of its forms (one broadcast per slot; point-to-point sends and receives): per form two sets of patterned
    records with ragged counts (a frame without features among them), every gathered byte and count compared with
    what the pattern says, then three timed rounds.  One line of JSON per form on stdout as soon as the form is
    through.  The parents take the point-to-point form where it worked on every rank and was not slower, else the
    broadcasts; if neither worked (or the child did not return within the parent's time limit) the measured run
    exchanges through torch.distributed's all-gather instead."""
    import datetime
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("SIFT_BENCH_PREFLIGHT_FAIL"):  # for tests of the parents' fallback
        sys.exit(5)
    import torch
    import torch.distributed as dist
    import sift_amd
    dev_index = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    def mark(what):  # the parent reports the last line when the child fails or does not return
        sys.stderr.write("preflight rank %d: %s\n" % (rank, what))
        sys.stderr.flush()

    mark("started on device %d" % dev_index)
    if world > 1:
        # a store of the children's own (under torch.distributed.run the environment tells a tcp:// rendezvous to look
        # for the launcher's store instead of starting one)
        store = dist.TCPStore("127.0.0.1", int(os.environ["SIFT_BENCH_PREFLIGHT_PORT"]), world, rank == 0,
                              timeout=datetime.timedelta(seconds=90))
        dist.init_process_group("gloo", store=store, rank=rank, world_size=world,
                                timeout=datetime.timedelta(seconds=90))
    mark("ranks met over gloo")
    ids = [sift_amd.api.comm_unique_id() if rank == 0 else None]
    if world > 1:
        dist.broadcast_object_list(ids, src=0)
    ctx = sift_amd.SiftGPU(max_width=64, max_height=64, max_batch=1, device=dev_index)
    mark("entering sg_comm_create")
    comm = ctx.comm_create(ids[0], rank, world)
    mark("communicator built")
    n_img, cap, words = 4, 65536, 144 // 8  # 38 MB of padded records per rank

    def run_form(form):
        """Two checked rounds and three timed ones of the all-gather-v in one of its forms; (ok, milliseconds)."""
        os.environ["SIFTGPU_ALLGATHERV"] = form  # read by the library at every call
        ok, best = True, None
        for trial in range(5):
            def counts_of(r):
                return [((r * 7919 + i * 104729 + trial * 31337 + 12345) % cap) if (r + i + trial) % 5 else 0
                        for i in range(n_img)]

            def tag(r):
                return (r << 48) | (trial << 40)

            mine = torch.arange(n_img * cap * words, dtype=torch.int64, device=dev) ^ tag(rank)
            counts = torch.tensor(counts_of(rank), dtype=torch.int32, device=dev)
            cap_all = world * n_img * cap
            g = torch.full((cap_all * words,), -1, dtype=torch.int64, device=dev)
            gc = torch.full((world * n_img,), -1, dtype=torch.int32, device=dev)
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
            mark("entering sg_allgather_features (%s), round %d" % (form, trial))
            t0 = time.perf_counter()
            first = ctx.allgather_features(comm, mine.data_ptr(), counts.data_ptr(), n_img, cap, g.data_ptr(), cap_all,
                                           gc.data_ptr(), world)
            ctx.sync()
            ms = 1e3 * (time.perf_counter() - t0)
            mark("gathered (%s), round %d" % (form, trial))
            if trial >= 2:  # the first rounds carry set-up costs of the communicator's channels
                best = ms if best is None else min(best, ms)
                continue
            ok = ok and gc.cpu().tolist() == [c for r in range(world) for c in counts_of(r)]
            at = 0
            for r in range(world):
                ok = ok and first[r] == at
                for i, c in enumerate(counts_of(r)):
                    want = torch.arange(i * cap * words, (i * cap + c) * words, dtype=torch.int64, device=dev) ^ tag(r)
                    ok = ok and torch.equal(g[at * words: (at + c) * words], want)
                    at += c
            ok = ok and first[world] == at and bool((g[at * words:] == -1).all().item())
        return ok, best

    # One form after the other, the result of each on a line of its own as soon as it is known: should the second form
    # never return, the parent (which ends this process at its time limit) still has the first one's line.  A watchdog
    # of this process's own ends a form that hangs after 45 s, so that a hang costs less than the parent's whole limit.
    import threading
    results = {}
    for form in ("bcast", "p2p"):
        dog = threading.Timer(45.0, lambda: os._exit(7))
        dog.daemon = True
        dog.start()
        ok, ms = run_form(form)
        dog.cancel()
        results[form] = ok
        sys.stdout.write(json.dumps({"form": form, "ok": bool(ok), "ms": ms}) + "\n")
        sys.stdout.flush()
        mark("form %s: %s" % (form, "all gathered bytes and counts as expected" if ok else "gathered bytes or counts differ"))
    ctx.comm_destroy(comm)
    ctx.close()
    if world > 1:
        dist.destroy_process_group()
    sys.exit(0 if results.get("bcast") else 3)


def preflight_abi_exchange(rank, world, dist, torch, dev):
    """Runs preflight_child() beside every rank and lets the ranks agree on the outcome.  Returns a dict for the
    bench line; ["ok"] says whether the measured run may use the C ABI's exchange."""
    import socket
    import subprocess
    import tempfile
    on_dev = dist.get_backend() == "nccl"
    port = [None]
    if rank == 0:
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port[0] = s.getsockname()[1]
        s.close()
    dist.broadcast_object_list(port, src=0)
    limit = float(os.environ.get("SIFT_BENCH_PREFLIGHT_TIMEOUT", "150"))
    t0 = time.perf_counter()
    why = ""
    forms = {}  # form -> (ok, ms) as this rank's child reported it
    with tempfile.TemporaryFile(mode="w+") as err, tempfile.TemporaryFile(mode="w+") as out:
        p = subprocess.Popen([sys.executable, os.path.abspath(__file__), "--preflight-child"], stdout=out,
                             stderr=err, env=dict({k: v for k, v in os.environ.items() if not k.startswith("TORCHELASTIC_")},
                                                  SIFT_BENCH_PREFLIGHT_PORT=str(port[0])))
        try:
            rc = p.wait(timeout=limit)
        except subprocess.TimeoutExpired:
            p.kill()
            p.wait()
            rc = None
        out.seek(0)
        for ln in out.read().splitlines():  # one line per form the child got through, whatever happened after it
            try:
                r = json.loads(ln)
                forms[r["form"]] = (bool(r["ok"]), float(r["ms"]) if r["ms"] is not None else 1e9)
            except (ValueError, KeyError, TypeError):
                pass
        if rc != 0:
            err.seek(0)
            lines = err.read().strip().splitlines()
            marks = [ln for ln in lines if ln.startswith("preflight rank")]
            errs = [ln for ln in lines if not ln.startswith(("preflight rank", "frame #", " ", "\t"))]
            why = (("no exit within %.0f s" % limit) if rc is None else "exit status %d" % rc) + \
                ("; got as far as: " + marks[-1][-120:] if marks else "") + ("; " + errs[-1][-200:] if errs else "")
    where = dev if on_dev else "cpu"
    oks = torch.tensor([1 if forms.get(f, (False, 0))[0] else 0 for f in ("bcast", "p2p")], dtype=torch.int32, device=where)
    dist.all_reduce(oks, op=dist.ReduceOp.MIN)  # a form counts when it worked on every rank
    ms = torch.tensor([forms.get(f, (False, 1e9))[1] for f in ("bcast", "p2p")], dtype=torch.float64, device=where)
    dist.all_reduce(ms, op=dist.ReduceOp.MAX)
    bcast_ok, p2p_ok = bool(oks[0].item()), bool(oks[1].item())
    bcast_ms, p2p_ms = float(ms[0].item()), float(ms[1].item())
    # the point-to-point form where it worked everywhere and was not slower; else the broadcasts; else torch.distributed
    form = "p2p" if p2p_ok and (not bcast_ok or p2p_ms <= bcast_ms) else ("bcast" if bcast_ok else None)
    return {"ok": form is not None, "form": form, "bcast": {"ok": bcast_ok, "ms": bcast_ms if bcast_ok else None},
            "p2p": {"ok": p2p_ok, "ms": p2p_ms if p2p_ok else None},
            "this_rank_ok": rc == 0, "this_rank": why or "ok", "seconds": time.perf_counter() - t0,
            "what": "trial all-gather-v through the C ABI's RCCL binding in child processes, in both of its forms (one "
                    "broadcast per slot; point-to-point sends and receives): per form two rounds of patterned records "
                    "with ragged counts, every byte compared, and three timed rounds of 38 MB per rank -- before the "
                    "measured processes set up theirs; ms = slowest rank's best round"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=256,
                    help="frames per step per GPU (the f64 pyramid of a 1080p frame takes 0.53 GB; halved until the\n"
                         "context fits when the device has less memory free)")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--precision", choices=("f64", "f32", "f64_fma"), default="f64",
                    help="f64: the reference's arithmetic (separate multiplies and adds), bit-identical pyramid;\n"
                         "f64_fma: double pyramid, fused multiply-adds in the Gaussian filters (tolerance mode);\n"
                         "f32: float pyramid (tolerance mode)")
    ap.add_argument("--pool", type=int, default=0,
                    help="distinct frames kept resident per GPU (0: 256 on one GPU, 512 per rank on several -- the\n"
                         "256-frame and the 4096-frames-over-8-GPUs configurations)")
    ap.add_argument("--exchange", action="store_true",
                    help="run the all-gather step on one rank too (a communicator of one; N > 1 always runs it)")
    ap.add_argument("--host-multi", action="store_true",
                    help="with several ranks: after the timed region rank 0 also runs the host C++ multi-GPU driver\n"
                         "(SiftBatchMultiGpu, one process, one thread per GPU) over the same number of GPUs and reports\n"
                         "its rate under host_api.multi_gpu_driver (on one GPU the extras always include it)")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the untimed extras (batch check, single-frame latencies, host-API throughput)")
    ap.add_argument("--cap", type=int, default=26624, help="packed-feature capacity per frame")
    ap.add_argument("--contexts", type=int, default=1,
                    help="batches in flight per GPU (1: per-kernel event times are undisturbed; 2 overlaps\n"
                         "memory-bound and compute-bound kernels of consecutive batches for a few % more throughput)")
    ap.add_argument("--cpu-threads", type=int, default=0, help="threads of the CPU baseline (0 = box share)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true", help="do not record per-launch HIP events")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    n_gpus = args.gpus
    if world != n_gpus and world > 1:
        n_gpus = world

    import torch
    import torch.distributed as dist
    import sift_amd
    from sift_amd import dist as sdist

    # The host-API programs (Sift() through the C++ layer, compiled programs of their own) are measured before this
    # process opens the GPU: a second process with queues on the device -- even an idle one -- makes the hardware
    # scheduler switch between the two, which a call that is a chain of fifty dependent launches feels (one Sift()
    # call 2.2 -> 2.5 ms, sixteen threads with a context each 870 -> 600 calls/s when measured from inside).
    host_api_early = None
    if rank == 0 and world == 1 and not args.no_extras:
        host_api_early = host_api_throughput(args.width, args.height, 0, args.cpu_threads)
        host_api_early["multi_gpu_driver"] = host_multi_gpu_driver(args.width, args.height, 1, 128)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback)")
    # One GPU per rank.  For a rehearsal of the N > 1 control flow on a box with fewer GPUs than ranks, ranks wrap
    # around the devices present and SIFT_BENCH_BACKEND=gloo replaces RCCL (which refuses two ranks on one GPU).
    n_dev = torch.cuda.device_count()
    dev_index = local_rank % max(n_dev, 1)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("SIFT_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    # The exchange step goes through the C ABI (sg_allgather_features) unless SIFT_BENCH_EXCHANGE=torch, a rehearsal
    # backend, or a failed trial of it (preflight_abi_exchange) sends it through torch.distributed.
    use_abi = world == 1 or (dist.get_backend() == "nccl" and os.environ.get("SIFT_BENCH_EXCHANGE", "abi") != "torch")
    preflight = None
    pf_mode = os.environ.get("SIFT_BENCH_PREFLIGHT", "1")
    if world > 1 and pf_mode != "0" and (use_abi or pf_mode == "force"):
        preflight = preflight_abi_exchange(rank, world, dist, torch, dev)
        use_abi = use_abi and preflight["ok"]
        if preflight["form"]:
            os.environ["SIFTGPU_ALLGATHERV"] = preflight["form"]  # the form the library's exchange takes from here on

    W, H, B, K, WU = args.width, args.height, args.batch, args.steps, args.warmup
    if args.pool <= 0:
        args.pool = 256 if world == 1 else 512
    prec = {"f64": sift_amd.SG_F64, "f32": sift_amd.SG_F32, "f64_fma": sift_amd.SG_F64_FMA}[args.precision]
    esize = 4 if prec == sift_amd.SG_F32 else 8
    nctx = max(1, min(args.contexts, 4))
    def buffers_bytes(b):
        """Device memory this script allocates beside the contexts: the resident frames, the packed-feature buffers
        and, with more than one rank, the gathered ones."""
        ex = world > 1 or args.exchange
        nb = max(2, nctx) if ex else nctx
        pool_frames = max(b, min(args.pool, (K + WU) * b))
        return pool_frames * W * H * 3 + nb * b * args.cap * 144 * (1 + (world if ex else 0)) + (1 << 30)

    ctxs = None
    while ctxs is None:
        try:
            ctxs = [sift_amd.SiftGPU(max_width=W, max_height=H, max_batch=B, precision=prec, device=dev_index)
                    for _ in range(nctx)]
            if torch.cuda.mem_get_info(dev)[0] < buffers_bytes(B) and B > 8:
                raise sift_amd.SiftGpuError(sift_amd.api.SG_E_NOMEM, "no room left for the frame and feature buffers")
        except sift_amd.SiftGpuError as e:  # not enough device memory for this many frames in flight
            if e.status != sift_amd.api.SG_E_NOMEM or B <= 8:
                raise
            for c in ctxs or []:
                c.close()
            ctxs = None
            B //= 2
    if world > 1:  # every rank takes the same number of frames per step: the smallest any rank could fit
        agreed = torch.tensor([B], dtype=torch.int64, device=dev if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(agreed, op=dist.ReduceOp.MIN)
        if int(agreed.item()) != B:
            B = int(agreed.item())
            for c in ctxs:
                c.close()
            ctxs = [sift_amd.SiftGPU(max_width=W, max_height=H, max_batch=B, precision=prec, device=dev_index)
                    for _ in range(nctx)]

    # ---- synthetic frames, generated on the device, resident in HBM ----
    pool = max(B, min(args.pool, (K + WU) * B))
    pool -= pool % B
    frame_bytes = W * H * 3
    frames = torch.empty(pool * frame_bytes, dtype=torch.uint8, device=dev)
    seed0 = 1000 + rank * pool  # global frame f = rank * pool + i has seed 1000 + f (f // 512 = rank at the default pool)
    for i in range(pool):
        ctxs[0].synth_vnoise5(frames.data_ptr() + i * frame_bytes, W, H, seed0 + i)
    ctxs[0].sync()

    cap = args.cap
    # Output buffers: one per context; with more than one rank at least two, so that the all-gather of one step's
    # features never reads a buffer the next step's extraction is writing.
    nbuf = nctx if not (world > 1 or args.exchange) else max(2, nctx)
    outs = [torch.empty(B * cap * 144, dtype=torch.uint8, device=dev) for _ in range(nbuf)]
    cnts = [torch.zeros(B, dtype=torch.int32, device=dev) for _ in range(nbuf)]
    exchange = xctx = comm = None
    do_exchange = world > 1 or args.exchange
    # With several ranks, the set-up of the C ABI's own RCCL communicator and the first all-gather through it (in the
    # warm-up) are the first time that path meets real peers.  Should either never return, the run ends with a message
    # instead of sitting until the caller's time limit: SIFT_BENCH_EXCHANGE=torch then takes the same exchange through
    # torch.distributed instead.
    watchdog = None
    if world > 1 and use_abi:
        import threading

        def stuck():
            sys.stderr.write("bench.py rank %d: the RCCL set-up or the warm-up exchange (sg_allgather_features) did not "
                             "finish in 300 s; rerun with SIFT_BENCH_EXCHANGE=torch to exchange through "
                             "torch.distributed\n" % rank)
            sys.stderr.flush()
            os._exit(4)

        watchdog = threading.Timer(300.0, stuck)
        watchdog.daemon = True
        watchdog.start()
    if do_exchange:
        # The exchange step: all-gather-v of the ranks' feature sets (sg_allgather_features: counts first, then one
        # grouped set of RCCL broadcasts into an unpadded buffer).  It runs on the stream of a context of its own, so
        # that it overlaps the next step's kernels.  Rehearsals over gloo (several ranks on one GPU, where RCCL cannot
        # run) take the same exchange through torch.distributed.
        cap_all = world * B * cap
        g_outs = [torch.empty(cap_all * 144, dtype=torch.uint8, device=dev) for _ in range(nbuf)]
        g_cnts = [torch.zeros(world * B, dtype=torch.int32, device=dev) for _ in range(nbuf)]
        if use_abi:
            # Every rank first checks on its own that the library can bind RCCL, and the ranks agree on the outcome
            # before any of them enters the collective set-up: a rank that cannot must not leave the others waiting.
            why = ""

            def all_ok(ok):
                if world == 1:
                    return ok
                t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
                dist.all_reduce(t, op=dist.ReduceOp.MIN)
                return bool(t.item())

            try:
                my_id = sift_amd.api.comm_unique_id()
                ok = True
            except Exception as e:
                ok, why = False, repr(e)
            if all_ok(ok):
                try:
                    xctx = sift_amd.SiftGPU(max_width=64, max_height=64, max_batch=1, device=dev_index)
                    ids = [my_id if rank == 0 else None]
                    if world > 1:
                        dist.broadcast_object_list(ids, src=0)
                    comm = xctx.comm_create(ids[0], rank, world)
                except Exception as e:
                    ok, why = False, repr(e)
                ok = all_ok(ok)
            if ok:
                exchange = "sg_allgather_features (RCCL all-gather-v, %s)" % (
                    "point-to-point sends and receives" if os.environ.get("SIFTGPU_ALLGATHERV") == "p2p"
                    else "one broadcast per slot")
            else:
                if xctx is not None:
                    xctx.close()
                xctx = comm = None
                if world == 1:
                    raise RuntimeError("the exchange step needs RCCL through the C ABI: " + why)
                exchange = ("torch.distributed all-gather-v over %s (RCCL through the C ABI was not usable on every "
                            "rank%s)" % (dist.get_backend(), ": " + why if why else ""))
        elif world > 1 and dist.get_backend() == "nccl":
            exchange = "torch.distributed all-gather-v over nccl (%s)" % (
                "the trial of the C ABI's exchange failed: see exchange_preflight" if preflight is not None
                and not preflight["ok"] else "SIFT_BENCH_EXCHANGE=torch")
        else:
            exchange = "torch.distributed all-gather-v (%s rehearsal)" % dist.get_backend()
    gathered_total = [0] * nbuf   # features in the gathered buffer b
    gathered_first = [None] * nbuf  # rank_first of the gathered buffer b (index of each rank's first feature)
    gathered_v = [None] * nbuf      # (features, counts) of buffer b when the exchange went through torch.distributed
    gather_busy = [False] * nbuf  # all-gather in flight on buffer b
    pending = [None] * nctx       # buffer that context c's batch in flight writes

    def wait_gather(b):
        """Buffer b's all-gather (if any) has completed when this returns."""
        if gather_busy[b]:
            if xctx is not None:
                xctx.sync()
            gather_busy[b] = False

    def retire(c):
        """Wait for context c's batch; hand its features to the all-gather."""
        b = pending[c]
        if b is None:
            return
        ctxs[c].sync()
        pending[c] = None
        if do_exchange:
            if xctx is not None:
                first = xctx.allgather_features(comm, outs[b].data_ptr(), cnts[b].data_ptr(), B, cap,
                                                g_outs[b].data_ptr(), cap_all, g_cnts[b].data_ptr(), world)
                gathered_total[b] = first[-1]
                gathered_first[b] = first
                gather_busy[b] = True
            else:
                on_dev = dist.get_backend() == "nccl"
                gv, gc, first = sdist.allgather_features_v(outs[b] if on_dev else outs[b].cpu(),
                                                           cnts[b] if on_dev else cnts[b].cpu(), cap)
                gathered_total[b] = first[-1]
                gathered_first[b] = first
                gathered_v[b] = (gv, gc)

    def submit(step):
        c, b = step % nctx, step % nbuf
        retire(c)
        wait_gather(b)
        first = (step * B) % pool
        ctxs[c].extract_batch_device(frames.data_ptr() + first * frame_bytes, B, W, H, 3,
                                     outs[b].data_ptr(), cap, cnts[b].data_ptr())
        pending[c] = b
        last_buf[0] = b

    last_buf = [0]

    def drain():
        for c in range(nctx):
            retire(c)
        for b in range(nbuf):
            wait_gather(b)
        torch.cuda.synchronize()

    def barrier():
        if world > 1:
            dist.barrier()

    # ---- warmup ----
    # (With several ranks the first all-gather through the C ABI's own RCCL binding happens here, under the watchdog
    # started in front of the communicator's set-up.)
    for s in range(max(WU, 1 if world > 1 else 0)):
        submit(s)
    drain()
    if watchdog is not None:
        watchdog.cancel()
    for c in ctxs:
        if not args.no_profile:
            c.set_profiling(True)
        c.reset_stage_times()

    # ---- timed region: exactly K steps ----
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in range(K):
        submit(WU + s)
    drain()
    torch.cuda.synchronize()
    barrier()
    t1 = time.perf_counter()
    elapsed = t1 - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- per-stage device times (HIP events on the contexts' streams, recorded inside the timed region) ----
    stage_ms = {}
    stage_launches = {}
    for c in ctxs:
        for k, (ms, ln) in c.stage_times().items():
            stage_ms[k] = stage_ms.get(k, 0.0) + ms
            stage_launches[k] = stage_launches.get(k, 0) + ln
        c.set_profiling(False)
    feat_counts = cnts[last_buf[0]].cpu().numpy()
    work = ctxs[(WU + K - 1) % nctx].read_work_counters(B)  # trips of the histogram loops in the last batch
    pair_mask = ctxs[0].last_pair_octaves()  # octaves whose levels 1 and 2 came from one launch
    tail_first = ctxs[0].last_tail_octave()  # octaves from here on came from the one pyramid-tail launch
    gathered_last = gathered_total[last_buf[0]] if do_exchange else None

    last_first = ((WU + K - 1) * B) % pool  # first frame of the last step's batch

    # ---- the exchange checks itself (outside the timed region, every rank) ----
    # A wrong slot offset or a slot that never arrived would still print a throughput number, so the last step's
    # gathered buffer is verified on every rank: (1) this rank's slot equals its own features, compacted by
    # sg_compact_features; (2) the gathered counts equal an all-gather of the ranks' counts through torch.distributed;
    # (3) a position-weighted 64-bit checksum of the whole gathered buffer is the same on every rank.  Valid because
    # a frame's features are a pure function of its pixels (reference sift/sift.cpp:1646-1707).
    exchange_check = None
    if do_exchange:
        b = last_buf[0]
        first = gathered_first[b]
        on_dev = world == 1 or dist.get_backend() == "nccl"
        if xctx is not None:
            g_feat, g_cnt = g_outs[b], g_cnts[b]
        else:
            g_feat, g_cnt = gathered_v[b][0], gathered_v[b][1].reshape(-1)
        total = int(first[-1])
        mine = int(first[rank + 1] - first[rank])
        local_cnt = cnts[b] if on_dev else cnts[b].cpu()
        # (1) own slot
        tmp = torch.empty(max(mine, 1) * 144, dtype=torch.uint8, device=dev)
        ctxs[0].compact_features(outs[b].data_ptr(), cnts[b].data_ptr(), B, cap, tmp.data_ptr())
        ctxs[0].sync()
        own = g_feat[first[rank] * 144: first[rank + 1] * 144]
        own_ok = mine == int(local_cnt.clamp(max=cap).sum().item()) and \
            torch.equal(own.to(tmp.device), tmp[: mine * 144])
        # (2) counts
        if world > 1:
            want_cnt = torch.empty(world * B, dtype=local_cnt.dtype, device=local_cnt.device)
            dist.all_gather_into_tensor(want_cnt, local_cnt.reshape(-1).contiguous())
        else:
            want_cnt = local_cnt.reshape(-1)
        counts_ok = torch.equal(want_cnt.cpu().to(torch.int64), g_cnt.reshape(-1).cpu().to(torch.int64))
        # (3) checksum over all ranks' features
        words = g_feat[: total * 144].view(torch.int64)
        cs, piece = 0, 1 << 24  # in pieces: the gathered set of eight ranks is gigabytes
        for w0 in range(0, words.numel(), piece):
            w = words[w0: w0 + piece]
            weights = (torch.arange(w.numel(), dtype=torch.int64, device=w.device) + w0) * 2654435761 + 1
            cs = (cs + int((w * weights).sum().item())) & 0xffffffffffffffff
        if cs >= 1 << 63:
            cs -= 1 << 64  # as a signed 64-bit value, so that it travels in an int64 tensor
        where = dev if on_dev else "cpu"
        lo = torch.tensor([cs, 1 if (own_ok and counts_ok) else 0], dtype=torch.int64, device=where)
        hi = torch.tensor([cs], dtype=torch.int64, device=where)
        if world > 1:
            dist.all_reduce(lo, op=dist.ReduceOp.MIN)
            dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        checksum_equal = int(lo[0].item()) == int(hi[0].item())  # smallest = largest: the same on every rank
        exchange_check = {"ok": bool(lo[1].item()) and checksum_equal, "ranks": world, "features_gathered": total,
                          "own_slot_equal": bool(own_ok), "counts_equal": bool(counts_ok),
                          "all_ranks_own_slot_and_counts_ok": bool(lo[1].item()), "checksum_equal": checksum_equal,
                          "checksum": "%016x" % (cs & 0xffffffffffffffff)}
        del tmp
    for c in ctxs:
        c.close()
    if xctx is not None:
        xctx.comm_destroy(comm)
        xctx.close()

    # ---- untimed extras (rank 0; the other ranks are done) ----
    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        if exchange_check is not None and not exchange_check["ok"]:
            sys.exit("bench.py rank %d: the gathered feature sets failed their check: %s" % (rank, json.dumps(exchange_check)))
        return
    verify = latency = host_api = None
    if not args.no_extras:
        verify, latency = batch_check_and_latency(sift_amd, torch, dev, dev_index, prec, W, H, cap, frames, frame_bytes,
                                                  last_first, B, outs[last_buf[0]], feat_counts)
        host_api = host_api_early or host_api_throughput(W, H, dev_index, args.cpu_threads)
        if host_api_early is None and (world == 1 or args.host_multi):
            host_api["multi_gpu_driver"] = host_multi_gpu_driver(W, H, world, 128 * world)
    overlap = f64_fma = None
    if not args.no_extras and args.precision == "f64" and (W, H) == (1920, 1080):
        try:
            f64_fma = fma_mode_agreement(sift_amd, torch, dev, dev_index, W, H, cap, frames, frame_bytes, last_first, B,
                                         outs[last_buf[0]], feat_counts)
        except Exception as e:  # a report, never a dependency of the measured path
            f64_fma = {"error": repr(e)}
    if not args.no_extras and world == 1 and nctx == 1:
        del frames, outs, cnts
        torch.cuda.empty_cache()
        overlap = two_batches_in_flight(args, B)
        if f64_fma is not None and "error" not in f64_fma:
            f64_fma.update(fma_mode_throughput(args, B))

    images = K * B * world
    value = images / elapsed
    px = octave_pixels(W, H)
    P = float(sum(px))
    N0 = float(px[0]) if px else 0.0
    taps = [11, 11, 13, 17, 21, 27]
    radius_name = {0: 5, 1: 5, 2: 6, 3: 8, 4: 10, 5: 13}
    tname = "double" if esize == 8 else "float"
    kflag = ",fma" if args.precision == "f64_fma" else ""
    blur_mode = os.environ.get("SIFTGPU_BLUR", "")
    fused = blur_mode != "2pass"

    def fused_name(ups):
        """The fused kernel the library launches (kernels_blur.hip): input rows staged by direct-to-LDS loads
        (blur_dma_k); the base blur, which forms its input by upsampling, and SIFTGPU_BLUR_DIRECT=1 take the
        register-staged form (blur_win_k); SIFTGPU_BLUR=ring the LDS-ring variant."""
        if blur_mode == "ring":
            return "blur_fused_k"
        return "blur_win_k" if ups or os.environ.get("SIFTGPU_BLUR_DIRECT") else "blur_dma_k"

    dims = octave_dims(W, H)
    # Algorithmic bytes of one blur launch: every pixel of the level read once and written once
    # (2 * sizeof(T) per pixel) -- per launch of the fused kernel, or per pass of the two-pass kernels.
    # With the fused kernel in use, planes not larger than the radius take the two-pass kernels; their time
    # is reported under blur_row<level> and they are left out of the fused kernel's bytes and launches.
    blur = {}
    tot_bytes = tot_ms = 0.0
    for kind in (("col",) if fused else ("row", "col")):
        for lvl in range(6):
            key = "blur_%s%d" % (kind, lvl)
            ms, ln = stage_ms.get(key, 0.0), stage_launches.get(key, 0)
            octs = dims[:1] if lvl == 0 else dims[:tail_first]  # the tail's octaves are counted under pyr_tail
            if fused:
                octs = [d for d in octs if min(d) > radius_name[lvl]]
            if fused and lvl in (1, 2):  # octaves whose levels 1 and 2 one launch made are counted under blur_pair
                octs = [d for o, d in enumerate(dims[:tail_first])
                        if min(d) > radius_name[lvl] and not (pair_mask >> o) & 1]
            nbytes = 2.0 * esize * float(sum(r * c for r, c in octs)) * B * K
            if fused and lvl == 3 and blur_mode in ("", "win"):
                # the level-3 launch also writes DownSample's output (reference sift.cpp:59-103): level 0 of the next
                # octave, every pixel once
                nbytes += esize * float(sum(dims[o + 1][0] * dims[o + 1][1] for o in range(min(tail_first, len(dims) - 1))
                                            if min(dims[o]) > radius_name[lvl] and dims[o + 1] != dims[o])) * B * K
            ups = fused and lvl == 0 and blur_mode in ("", "win")
            if ups:  # the base blur reads the double gray plane at input resolution instead of an upsampled plane
                nbytes = (esize * float(dims[0][0] * dims[0][1]) + 8.0 * W * H) * B * K
            if ms > 0:
                kname = ("%s<%s,R=%d%s%s>" % (fused_name(ups), tname, radius_name[lvl],
                                               ",upsampling" if ups else "", kflag)) if fused else \
                    ("blur_%s_k<%s,R=%d>" % (kind, tname, radius_name[lvl]))
                blur[key] = {"kernel": kname, "taps": taps[lvl], "launches": int(ln), "avg_us": 1e3 * ms / max(ln, 1),
                             "GBps": nbytes / (ms * 1e-3) / 1e9, "bytes_per_launch_avg": nbytes / max(ln, 1)}
                tot_bytes += nbytes
                tot_ms += ms
    # Levels 1 and 2 of an octave from one launch (blur_pair_k): level 0 read once, both levels written once, i.e.
    # 3 * sizeof(T) per pixel of the octave and launch (the two single-level launches it replaces move 4).
    pair_bytes_saved = 0.0  # what the same levels would move as single-level launches, beyond what the pairs move
    for key, mask, radii, ntap in (("blur_pair", pair_mask, "5+6", 11),):
        ms, ln = stage_ms.get(key, 0.0), stage_launches.get(key, 0)
        if ms > 0:
            nbytes = 3.0 * esize * float(sum(r * c for o, (r, c) in enumerate(dims) if (mask >> o) & 1)) * B * K
            blur[key] = {"kernel": "blur_pair_k<%s,R=%s>" % (tname, radii), "taps": ntap, "launches": int(ln),
                         "avg_us": 1e3 * ms / max(ln, 1), "GBps": nbytes / (ms * 1e-3) / 1e9,
                         "bytes_per_launch_avg": nbytes / max(ln, 1),
                         "GBps_as_two_launches": nbytes * (4.0 / 3.0) / (ms * 1e-3) / 1e9}
            tot_bytes += nbytes
            tot_ms += ms
            pair_bytes_saved += nbytes / 3.0
    # The smallest octaves from one launch (pyr_tail_k: one workgroup per frame, planes in LDS): level 0 of its first
    # octave read once, five levels of every octave and level 0 of the later ones written once.
    ms, ln = stage_ms.get("pyr_tail", 0.0), stage_launches.get("pyr_tail", 0)
    if ms > 0:
        tail_px = [r * c for r, c in dims[tail_first:]]
        nbytes = esize * float(tail_px[0] + 5 * sum(tail_px) + sum(tail_px[1:])) * B * K
        blur["pyr_tail"] = {"kernel": "pyr_tail_k<%s> (octaves %d..%d)" % (tname, tail_first, len(dims) - 1), "taps": 27,
                            "launches": int(ln), "avg_us": 1e3 * ms / max(ln, 1), "GBps": nbytes / (ms * 1e-3) / 1e9,
                            "bytes_per_launch_avg": nbytes / max(ln, 1)}
        tot_bytes += nbytes
        tot_ms += ms
    # the kernel name the profiler reports merges stage slots that share a template instantiation (R=5: slots 0, 1)
    by_kernel = {}
    for key, d in blur.items():
        e = by_kernel.setdefault(d["kernel"], {"ms": 0.0, "launches": 0, "bytes": 0.0, "taps": d["taps"]})
        e["ms"] += d["avg_us"] * d["launches"] * 1e-3
        e["launches"] += d["launches"]
        e["bytes"] += d["bytes_per_launch_avg"] * d["launches"]
    det_ms = stage_ms.get("detect", 0.0)
    detect = None
    if det_ms > 0:
        det_bytes = 6.0 * esize * P * B * K  # the six Gaussian levels of every octave, read once
        det_ln = int(stage_launches.get("detect", 0))
        detect = {"kernel": "detect_k<%s> (fused DoG + 3x3x3 extrema scan; one launch over all octaves)" % tname,
                  "launches": det_ln, "GBps": det_bytes / (det_ms * 1e-3) / 1e9,
                  "frac": det_bytes / (det_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS}
        by_kernel["detect_k<%s>" % tname] = {"ms": det_ms, "launches": det_ln, "bytes": det_bytes, "taps": 0}
    # roofline: the HBM-side (pyramid) kernel with the largest share of the timed region -- a blur instantiation or the
    # fused DoG + extrema scan.  The histogram kernels (orientation, descriptors) are VALU/LDS-atomic work and are
    # reported as time only.
    roofline = None
    if by_kernel:
        dom = max(by_kernel, key=lambda k: by_kernel[k]["ms"])
        e = by_kernel[dom]
        gbps = e["bytes"] / (e["ms"] * 1e-3) / 1e9
        if dom.startswith("detect_k"):
            what = " (six Gaussian levels -> DoG -> 3x3x3 extrema bitmap; all octaves' launches)"
        elif fused:
            what = " (row + column pass of a %d-tap Gaussian in one launch; all octaves' launches)" % e["taps"]
        else:
            what = " (%d-tap pass; all octaves' launches)" % e["taps"]
        roofline = {
            "bound": "hbm", "kernel": dom + what,
            "achieved": gbps, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": gbps / HBM_PEAK_GBPS,
            "traffic": measured_traffic(dom, tname, B, W, H),
            "launches": e["launches"], "avg_launch_us": 1e3 * e["ms"] / e["launches"],
            "algorithmic_bytes_per_launch": e["bytes"] / e["launches"],
            "measured_stream_ceilings_GBps": measured_stream_ceilings(),
            "all_blur_launches_GBps": (tot_bytes / (tot_ms * 1e-3) / 1e9) if tot_ms > 0 else None,
            # the same launches priced as in a pyramid that blurs level by level (2 * sizeof(T) per pixel and level):
            # the figure to compare with a run that has no two-level launch
            "all_blur_GBps_level_by_level_bytes": ((tot_bytes + pair_bytes_saved) / (tot_ms * 1e-3) / 1e9)
            if tot_ms > 0 else None,
        }

    cpu_baseline = None
    if not args.no_cpu_baseline and world == 1:
        try:
            from oracle import oracle as O  # checker only: times the CPU restatement beside the GPU path
            threads = args.cpu_threads or min(16, os.cpu_count() or 1)
            imgs = np.stack([O.vnoise5(W, H, 1000 + i) for i in range(threads)])
            sec, counts = O.sift_batch(imgs, threads)
            cpu_baseline = {"value": threads / sec, "unit": "images/s", "cores": threads, "kind": "port",
                            "sample": "%d frames %dx%d (seeds 1000..%d), one frame per thread, %.1f s wall; "
                                      "scalar double-precision C restatement of the reference (not OpenCV-accelerated)"
                                      % (threads, W, H, 999 + threads, sec),
                            "features_frame0": int(counts[0])}
        except Exception as e:  # the baseline is a report, never a dependency of the measured path
            cpu_baseline = {"error": repr(e)}

    line = {
        "metric": "1080p images/sec/GPU (keypoints+descriptors)",
        "value": value, "unit": "images/s", "n_gpus": world, "steps": K, "warmup": WU,
        "ms_per_step": 1e3 * elapsed / K, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64" if esize == 8 else "f32", "data": "synthetic",
        "config": {"workload": "%d frames %dx%d per GPU (%d steps x %d frames; %d distinct frames, vnoise5 seeds "
                               "%d.., taken in turn), sigma 1.6, 3 intervals, double-size base, packed 144-B features%s"
                               % (K * B, W, H, K, B, pool, seed0,
                                  (", %d distinct frames per rank (global frame f: seed 1000 + f, rank f // %d), "
                                   "all-gather-v of the feature sets after every step" % (pool, pool)) if world > 1 else ""),
                   "batch": B, "contexts_in_flight": nctx, "precision": args.precision,
                   "features_per_frame_last_batch": [int(x) for x in feat_counts[:4]]},
        "images_per_sec_per_gpu": value / world,
        "roofline": roofline, "cpu_baseline": cpu_baseline,
        "valu_roofline": [r for r in (
            valu_roofline("descr_k<double,false>", work[0], stage_ms.get("descr", 0.0) / K) if esize == 8 else None,
            valu_roofline("orient_k<double,false>", work[1], stage_ms.get("orient", 0.0) / K) if esize == 8 else None) if r],
        "batch_check": verify, "latency_ms": latency, "host_api": host_api,
        "two_batches_in_flight": overlap, "f64_fma": f64_fma, "exchange": exchange, "exchange_preflight": preflight,
        "exchange_check": exchange_check,
        "features_last_step": int(feat_counts.astype(np.int64).sum()), "features_gathered_last_step": gathered_last,
        "stage_ms_per_step": {k: v / K for k, v in stage_ms.items() if v > 0},
        "blur_kernels": blur, "detect_kernel": detect,
    }
    print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()
    if exchange_check is not None and not exchange_check["ok"]:
        sys.exit("bench.py: the gathered feature sets failed their check: %s" % json.dumps(exchange_check))


if __name__ == "__main__":
    if "--preflight-child" in sys.argv[1:]:
        preflight_child()
    main()
