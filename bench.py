#!/usr/bin/env python3
"""bench.py — ORB extract + stereo-match throughput on MI355X (BASELINE.json metric, config C3).

A step = one pass of the hot path over one batch of B synthetic 1280x720 rectified stereo pairs per GPU:
both-eye ORBextractor::operator() (pyramid, per-cell FAST, quadtree, orientation, blur, rBRIEF) followed by
Frame::ComputeStereoMatches, all in the hand-written HIP kernels of liborbx.so, inputs resident in HBM.
One process per GPU; frames are independent, so ranks share nothing on the data path ("scaling": "weak");
torch.distributed (RCCL) is used for the barriers and the max-over-ranks time only.

Prints ONE JSON line (rank 0) with the driver's contract fields plus
  "roofline":     dominant kernel, algorithmic bytes / HIP-event time vs the 8 TB/s HBM peak
  "cpu_baseline": the CPU oracle (single-thread port of the reference's serial semantics) on a bounded sample
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # 100 steps = 0.1 s of GPU time: with two batches in flight the first and the last step have no partner to overlap
    # with, which costs a 20-step run ~5 % (33.3 k vs 35.0 k pairs/s at 100 steps, 35.25 k at 300)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--pairs", type=int, default=32, help="stereo pairs per step per GPU")
    ap.add_argument("--width", type=int, default=1280)
    ap.add_argument("--height", type=int, default=720)
    ap.add_argument("--nfeatures", type=int, default=1500)
    ap.add_argument("--distinct", type=int, default=4, help="distinct synthetic streams generated per GPU")
    ap.add_argument("--cpu-pairs", type=int, default=40, help="pairs timed on the CPU oracle (0 = skip)")
    ap.add_argument("--cpu-cores", type=int, default=0, help="worker processes of the CPU baseline (0 = all cores)")
    ap.add_argument("--no-profile", action="store_true", help="do not bracket kernels with HIP events")
    ap.add_argument("--handles", type=int, default=2,
                    help="extractor handles used round-robin (each owns streams + buffers); 2 = double buffering: the "
                         "next batch's pyramid / FAST overlaps the tail of the previous one (+12%% over 1)")
    ap.add_argument("--mode", choices=("stereo", "mono", "fisheye"), default="stereo",
                    help="stereo = BASELINE config C3 (the headline metric); mono = extraction only (C2: --width 640 "
                         "--height 480 --nfeatures 1000), value counts single frames; fisheye = C4 (--width 512 --height 512):"
                         " lapping areas + ComputeStereoFishEyeMatches (2-NN + KB8 triangulation) on the device")
    ap.add_argument("--allgather", action="store_true",
                    help="config C5 extra: RCCL all-gather of every rank's descriptor blocks after each step")
    return ap.parse_args()


def usable_cores():
    """CPU cores this process may actually use: affinity mask capped by the cgroup CPU quota (a container on a
    256-core host is often limited to a handful of CPUs, and os.cpu_count() does not know)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, int(float(q) / float(p) + 0.5)))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, int(q / p + 0.5)))
        except Exception:
            pass
    return max(1, n)


def algorithmic_bytes(stage, NI, P, plevels, ncand, nsel, nmatch_in, npairs):
    """Algorithmic HBM bytes of one batch for each kernel (SURVEY.md 8d per-unit figures x units/launch).
    NI images, P = sum of level pixels, plevels = pixels per level, ncand / nsel = mean candidates /
    selected keypoints per image, npairs stereo pairs."""
    if stage == "k_resize":      # level l reads level l-1 and writes level l
        return NI * sum(plevels[l - 1] + plevels[l] for l in range(1, len(plevels)))
    if stage == "k_detect":      # one compulsory read of every level + 4 B per emitted candidate
        return NI * (P + 4 * ncand)
    if stage == "k_octree":      # candidates in (4 B), selected keypoints out (4 B)
        return NI * (4 * ncand + 4 * nsel)
    if stage == "k_blur":        # read P, write P
        return NI * 2 * P
    if stage == "k_slots":
        return NI * 8 * nsel
    if stage == "k_describe":    # 749 B patch (angle) + 37x37 blurred footprint + 28 B keypoint + 32 B descriptor
        return NI * nsel * (749 + 1369 + 28 + 32)
    if stage == "k_stereo_match":  # (28+32) B per keypoint of both eyes + 352 B SAD windows per matched keypoint
        return npairs * (60 * 2 * nsel + 352 * nmatch_in)
    if stage == "k_stereo_filter":
        return npairs * 12 * nsel
    return 0


def main():
    a = parse()
    # Exactly ONE line on stdout: libraries (RCCL prints a version banner when NCCL_DEBUG=VERSION is set, HIP /
    # libdrm print warnings) must not interleave with it, so fd 1 is pointed at stderr for the run and the JSON
    # line is written to the saved descriptor at the end.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus and world > 1:
        a.gpus = world

    import numpy as np
    import torch  # first: liborbx.so then binds to the same HIP runtime as torch (SONAME libamdhip64.so.7)

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the ORB front-end has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or os.environ.get("ORBX_FORCE_DIST") == "1":  # the env switch lets a 1-GPU box exercise the RCCL path
        import torch.distributed as dist_
        dist = dist_
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    import orb_slam3_fast_amd as orbx
    from orb_slam3_fast_amd import sharding, synth

    W, H, B, NF = a.width, a.height, a.pairs, a.nfeatures
    # ---- synthetic streams (deterministic): D distinct pairs tiled to B per GPU
    D = max(1, min(a.distinct, B))
    pairs = [synth.stereo_pair(W, H, stream=1000 * rank + i, frame=0) for i in range(D)]
    lefts = np.stack([pairs[i % D][0] for i in range(B)])
    rights = np.stack([pairs[i % D][1] for i in range(B)])
    images = torch.from_numpy(np.concatenate([lefts, rights])).cuda(local_rank)  # [2B, H, W]: L0..LB-1 R0..RB-1
    torch.cuda.synchronize()

    exs = [orbx.ORBextractor(NF, 1.2, 8, 20, 7, max_width=W, max_height=H, max_batch=2 * B, device=local_rank)
           for _ in range(max(1, a.handles))]
    ex = exs[0]
    step_no = [0]
    bf, b = 0.12 * 532.03, 0.12  # ZED2-like rig: fx = 532.03 px, baseline 0.12 m
    ptr = images.data_ptr()

    gather_out = None

    class _Raw:  # zero-copy view of a liborbx device buffer as a torch tensor
        def __init__(self, ptr, shape, typestr):
            self.__cuda_array_interface__ = {"shape": shape, "typestr": typestr, "data": (ptr, False), "version": 2}

    lap = None
    if a.mode == "fisheye":  # TUM-VI-like lapping areas (Examples/Stereo-Inertial/TUM-VI.yaml:45-49 scaled to W)
        lap = np.array([[W // 5, W - 1]] * B + [[0, (4 * W) // 5]] * B, np.int32)
        rig = orbx.kb8_rig(synth.TUMVI_CAM1, synth.TUMVI_CAM2, np.eye(3), [0.101, 0.002, 0.001])

    def step():
        ex = exs[step_no[0] % len(exs)]
        step_no[0] += 1
        ex.extract_batch_device(ptr, 2 * B, W, H, W, W * H, lap=lap)
        if a.mode == "stereo":
            orbx.stereo_match_async(ex, ex, bf, b, first_left=0, first_right=B, n_pairs=B)
        elif a.mode == "fisheye":
            orbx.fisheye_match_async(ex, ex, rig, first_left=0, first_right=B, n_pairs=B)
        if a.allgather and dist is not None:
            # config C5 extra: every GPU ends up with all cameras' descriptor blocks (RCCL all-gather over xGMI)
            ex.sync()
            d_kps, d_desc, d_cnt, d_mono, cap = ex.results_device()
            desc = torch.as_tensor(_Raw(d_desc, (2 * B, cap, 32), "|u1"), device="cuda")
            cnt = torch.as_tensor(_Raw(d_cnt, (2 * B,), "<i4"), device="cuda")
            nonlocal gather_out
            gather_out = sharding.allgather_descriptor_blocks(cnt, desc, cap)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def collect():
        acc = {}
        for e in exs:
            for k, v in e.profile_collect().items():
                acc[k] = (acc.get(k, (0.0, 0))[0] + v[0], acc.get(k, (0.0, 0))[1] + v[1])
        return acc

    # warm-up: every kernel bracketed with HIP events -> which kernel dominates
    for e in exs:
        e.profile_enable(not a.no_profile)
    for _ in range(max(a.warmup, len(exs))):
        step()
    barrier()
    dom = None
    if not a.no_profile:
        wprof = collect()
        dom = max(wprof, key=lambda k: wprof[k][0])
        for e in exs:
            e.profile_enable(True, stage=dom)   # timed region: only the dominant kernel is bracketed
    barrier()
    # The host side of a step is ~60 us of Python; a generation-2 garbage collection (tens of ms with torch's object graph
    # loaded) that happens to fall into the 20 timed steps would be billed to the GPU path -- it made C4 read 13 k instead
    # of 63 k pairs/s in every process but the first one on a box.  Collect now, then keep the collector out of the region.
    import gc
    gc.collect()
    gc.disable()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    barrier()
    t1 = time.perf_counter()
    gc.enable()
    elapsed = t1 - t0
    if dist is not None:
        elapsed = sharding.max_over_ranks(elapsed, device="cuda")
    dom_prof = collect() if not a.no_profile else {}
    # per-stage table: a short extra pass outside the timed region with every kernel bracketed, on ONE handle with a
    # sync after every step, so that the durations are those of the kernels alone (in the timed region the batches
    # of the two handles overlap on the GPU, which stretches every individual launch)
    prof = {}
    if not a.no_profile:
        exs[0].profile_enable(True)
        nprof = max(3, min(a.steps, 10))
        step_no[0] = 0
        for _ in range(nprof):
            step_no[0] = 0
            step()
            exs[0].sync()
        barrier()
        prof = collect()
        for e in exs:
            e.profile_enable(False)

    # ---- workload statistics for the algorithmic byte counts
    lw, lh, nc, ns = ex.level_stats(0)
    plevels = [int(x) * int(y) for x, y in zip(lw, lh)]
    P = sum(plevels)
    ncand_mean = float(np.mean([ex.level_stats(i)[2].sum() for i in range(0, 2 * B, max(1, 2 * B // 8))]))
    nsel_mean = float(np.mean([ex.level_stats(i)[3].sum() for i in range(0, 2 * B, max(1, 2 * B // 8))]))
    if a.mode == "stereo":
        d_u = np.zeros((1, ex.capacity), np.float32)
        orbx._check(orbx.lib().orbx_stereo_download(ex._h, 0, orbx._p(d_u[0]), None, ex.capacity))
        nmatch = int((d_u >= 0).sum())
    elif a.mode == "fisheye":
        nmatch = orbx.fisheye_download(ex, ex, 0)[0]
    else:
        nmatch = 0

    units_per_step = 2 * B if a.mode == "mono" else B  # mono: every image is a frame
    value = a.gpus * units_per_step * a.steps / elapsed
    out = {
        "metric": {"stereo": "ORB extract+match frames/sec @%d\u00d7%d stereo (both-eye ORBextractor + ComputeStereoMatches)",
                   "mono": "ORB extract mono frames/sec @%dx%d (ORBextractor::operator())",
                   "fisheye": "ORB extract+match fisheye stereo frames/sec @%dx%d (both-eye ORBextractor with lapping areas "
                              "+ ComputeStereoFishEyeMatches)"}[a.mode] % (W, H),
        "value": round(value, 2),
        "unit": "frames/s" if a.mode == "mono" else "stereo frames/s",
        "n_gpus": a.gpus,
        "steps": a.steps,
        "warmup": a.warmup,
        "ms_per_step": round(1000.0 * elapsed / a.steps, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u8",
        "data": "synthetic",
        "config": {
            "workload": {"stereo": "C3: synthetic %dx%d rectified stereo pairs, %d features, 8 levels, scale 1.2, FAST 20/7, "
                                   "ComputeStereoMatches (bf=0.12*532.03, b=0.12)",
                         "mono": "C2: synthetic %dx%d mono frames, %d features, 8 levels, scale 1.2, FAST 20/7",
                         "fisheye": "C4: synthetic %dx%d fisheye stereo pairs, %d features, 8 levels, scale 1.2, FAST 20/7, "
                                    "lapping areas, BF 2-NN + KannalaBrandt8 triangulation"}[a.mode] % (W, H, NF),
            "pairs_per_step_per_gpu": B,
            "handles": len(exs),
            "distinct_streams_per_gpu": D,
            "keypoints_per_image": round(nsel_mean, 1),
            "fast_candidates_per_image": round(ncand_mean, 1),
            "stereo_matches_pair0": nmatch,
            "parallelism": "independent stereo pairs sharded across GPUs, no data-path collective"
                           + (" + RCCL all-gather of descriptor blocks" if a.allgather and dist else ""),
        },
    }

    if prof:
        nprof = max(3, min(a.steps, 10))
        tot = sum(v[0] for v in prof.values())
        stages = {}
        for name, (ms, cnt) in prof.items():
            if cnt == 0:
                continue
            nb = algorithmic_bytes(name, 2 * B, P, plevels, ncand_mean, nsel_mean, nmatch, B) * nprof
            stages[name] = {"ms_total": round(ms, 3), "launches": cnt, "avg_us": round(1000.0 * ms / cnt, 2),
                            "share": round(ms / tot, 4), "algorithmic_GBps": round(nb / (ms * 1e-3) / 1e9, 1)}
        # roofline of the dominant kernel from the HIP events recorded IN the timed region
        ms, cnt = dom_prof[dom]
        per_launch = algorithmic_bytes(dom, 2 * B, P, plevels, ncand_mean, nsel_mean, nmatch, B) / max(1, cnt // a.steps)
        ach = per_launch / (ms / cnt * 1e-3) / 1e9
        traffic = None
        tfile = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tfile):  # HBM bytes per launch from rocprofv3 --pmc passes (see profiles/README.md)
            try:
                traffic = json.load(open(tfile)).get(dom, {}).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out["roofline"] = {"kernel": dom, "bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": round(ach / HBM_PEAK_GBS, 5), "traffic": traffic,
                           "avg_launch_us": round(1000.0 * ms / cnt, 2), "launches_timed": cnt,
                           "algorithmic_bytes_per_launch": int(per_launch),
                           "note": "k_detect is VALU-issue bound (PMC: ~80%% of the VALU issue slots), not HBM bound; with "
                                   "%d handles in flight the launches of consecutive batches overlap, so avg_launch_us is "
                                   "the duration under overlap; isolated_* is the same kernel alone (stage pass); see "
                                   "DESIGN.md 5" % len(exs)}
        if dom in stages:
            iso = stages[dom]["avg_us"]
            out["roofline"]["isolated_avg_launch_us"] = iso
            out["roofline"]["isolated_frac"] = round(per_launch / (iso * 1e-6) / 1e9 / HBM_PEAK_GBS, 5)
        out["stages"] = stages
        out["stages_note"] = ("per-stage HIP-event table from %d extra single-handle steps (synchronised, no overlap "
                              "between batches) after the timed region" % nprof)
        a_pair = 2 * (2 * P + 60 * nsel_mean) + 120 * nsel_mean + 352 * nmatch
        out["end_to_end_algorithmic_GBps"] = round(a_pair * value / a.gpus / 1e9, 2)

    # ---- CPU baseline: the oracle (port of the reference's serial semantics), rank 0, N=1 only.
    # Frame-parallel over the host cores in a separate process tree (`cpu_mt` of BASELINE.md), plus 1 core.
    if rank == 0 and a.gpus == 1 and a.cpu_pairs > 0 and a.mode == "stereo":
        import subprocess
        import tempfile
        tmp = os.path.join(tempfile.gettempdir(), "orbx_cpu_pairs_%d.npy" % os.getpid())
        np.save(tmp, np.stack([np.stack(p) for p in pairs]))
        cores = usable_cores() if a.cpu_cores <= 0 else a.cpu_cores
        per = max(2, a.cpu_pairs // 8)  # ~ per * 0.3 s per worker
        try:
            r1 = json.loads(subprocess.run([sys.executable, "-m", "oracle.cpu_bench", tmp, str(NF), str(bf), str(b), "1",
                                            str(max(4, a.cpu_pairs // 4))], cwd=ROOT, capture_output=True, text=True,
                                           timeout=300).stdout.strip().splitlines()[-1])
            rN = json.loads(subprocess.run([sys.executable, "-m", "oracle.cpu_bench", tmp, str(NF), str(bf), str(b),
                                            str(cores), str(per)], cwd=ROOT, capture_output=True, text=True,
                                           timeout=600).stdout.strip().splitlines()[-1])
            out["cpu_baseline"] = {
                "value": round(rN["pairs_per_s"], 2), "unit": "stereo frames/s", "cores": cores, "kind": "port",
                "single_core_value": round(r1["pairs_per_s"], 3),
                "sample": "oracle/liborb_oracle.so (g++ -O2 port of the reference's serial semantics), frame-parallel: "
                          "%d worker processes x %d of the same synthetic %dx%d pairs, wall %.1f s; single worker: %d pairs "
                          "in %.1f s; host reports %d cores, %d usable (affinity / cgroup quota)" % (
                              cores, per, W, H, rN["wall_s"], r1["pairs"], r1["wall_s"], os.cpu_count(), usable_cores())}
        finally:
            if os.path.exists(tmp):
                os.remove(tmp)

    if rank == 0:
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
