#!/usr/bin/env python3
"""Headline benchmark: decoded Mpixels/s at 1920x1080 (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W [--workload NAME] [--also a,b | --no-also]

Workload at every N (weak scaling): each rank owns its own independent stream(s) — by default
BASELINE.json configs[1], "MSVideo1 1920x1080 keyframe-only": 512 distinct frames of block mix M1 (25 % solid /
50 % 2-colour / 25 % 8-colour) whose RAW STREAM BYTES are resident in HBM.  A "step" = one pass of
the hot path over them: on-GPU parse + 4x4 block reconstruction into 512 distinct RGB32 frame buffers, through the
C ABI (jsp_staged_decode).  A step reads 531 MB of stream and writes 4.2 GB of frames, so nothing a step touches
is left in the 256 MiB Infinity Cache by the step before it.  Streams shard one per GPU; the only collective is
the counter reduce.

After the timed region every frame the timed kernels left in HBM is compared with the CPU oracle's digest of
the same frame (tests/golden/bench_digests.json, written by tests/golden/make_bench_digests.py): no value is
printed on a mismatch.  A second leg times the SAME frames end to end — compressed bytes in host memory -> host
stage -> H2D -> kernels, one frame per call through the asynchronous entry points — and reports it beside the
resident-input number ("e2e").

The metric names both codecs, so the default line carries the ScreenPressor configurations as well: `also` = one entry
per further workload (BASELINE.json configs[2] and [3]: ScreenPressor key frames, ScreenPressor 300-frame clips), each
timed, digest-verified and priced against the roofline exactly like the headline (same K, W, barriers, HIP events), with
a bounded CPU baseline and the many-stream end-to-end rate beside it.

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline`, `cpu_baseline`, `e2e`,
`verified`, `also`.  The oracle (oracle/) is used ONLY in the cpu_baseline legs.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
METRIC = "decoded Mpixels/sec at 1920x1080 (MSVideo1 + ScreenPressor), 1/2/4/8 GPUs"
# what the default line carries beside the headline workload (BASELINE.json configs[2], configs[3])
ALSO_DEFAULT = ("screenpressor_v4_1080p_iframes", "screenpressor_v4_1080p_pclip300")


def _oracle_stream(spec, clip, budget_s, max_frames, gate=None):
    """One oracle instance decoding the clip over and over for ~budget_s: (frames done, seconds).
    With `gate` (a threading.Barrier) the frame buffers are touched first and all streams start together
    (first-touch page faults of 8 threads at once would otherwise dominate a short sample)."""
    import numpy as np
    from oracle_binding import OracleMSVideo1, OracleScreenPressor
    from jsplayer_amd.workloads import W, H
    orc = OracleScreenPressor(W, H, 24) if spec["codec"] == "sp" else OracleMSVideo1(spec["bits"], W, H, clip.palette)
    orc.Preinit(36)
    bufs = [np.ones(W * H, dtype=np.int32) for _ in range(2)]
    if gate is not None:
        gate.wait()
    done, t0 = 0, time.perf_counter()
    while True:
        for src, key in zip(clip.frames, clip.keys):
            dst = bufs[0] if orc.PreviousFrame() is bufs[1] else bufs[1]
            if key:
                orc.DecompressI(src, dst)
            else:
                orc.DecompressP(src, dst)
            done += 1
            if (done & 15) == 0 and time.perf_counter() - t0 >= budget_s:
                return done, time.perf_counter() - t0
        el = time.perf_counter() - t0
        if el >= budget_s or done >= max_frames:
            return done, el


def cpu_baseline(spec, clip, budget_s=8.0):
    """Oracle (C++ restatement of the Haxe reference, -O2) on the same frames: one thread, then one
    independent stream per host core (the reference's only way to use more cores)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import threading
    from concurrent.futures import ThreadPoolExecutor
    from jsplayer_amd.workloads import W, H
    done, el = _oracle_stream(spec, clip, budget_s, 1 << 20)
    ncores = max(1, min(os.cpu_count() or 1, 16))
    gate = threading.Barrier(ncores + 1)
    with ThreadPoolExecutor(ncores) as ex:      # the ctypes calls release the GIL
        futs = [ex.submit(_oracle_stream, spec, clip, budget_s / 2, 1 << 20, gate) for _ in range(ncores)]
        gate.wait()
        t0 = time.perf_counter()
        parts = [f.result() for f in futs]
    wall = time.perf_counter() - t0
    out = {
        "all_cores": {"value": round(sum(d for d, _ in parts) * W * H / wall / 1e6, 2), "unit": "Mpixels/s",
                      "cores": ncores, "sample": f"{ncores} independent streams, one thread each, {wall:.1f} s"},
        "value": round(done * W * H / el / 1e6, 2),
        "unit": "Mpixels/s",
        "cores": 1,
        "kind": "port",
        "sample": f"{done} frames of the workload's first clip ({len(clip.frames)} distinct 1920x1080 frames, in order), "
                  f"{el:.1f} s, oracle/ C++ restatement -O2 single thread, DecompressI/P only (parse / entropy decode included)",
        "host_cores_available": os.cpu_count(),
    }
    try:
        with open("/proc/cpuinfo") as f:
            out["cpu_model"] = next(l.split(":", 1)[1].strip() for l in f if l.startswith("model name"))
    except Exception:
        pass
    return out


def _device_identity(index: int):
    """Which GPU this is (the boxes of the pool differ in what their memory makes of the decode kernels' store shapes: DESIGN.md 6)."""
    try:
        import torch
        p = torch.cuda.get_device_properties(index)
        return {"name": p.name, "uuid": str(getattr(p, "uuid", "")), "gcn_arch": getattr(p, "gcnArchName", ""), "total_memory_GB": round(p.total_memory / 1e9, 1)}
    except Exception:
        return None


def _spawn_ranks(n: int) -> int:
    """`python bench.py --gpus N` without a launcher: start `torch.distributed.run` with N ranks of this very command as a
    CHILD process — before anything has touched the GPU — and hand its output and exit code through."""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: what RCCL needs on this pool
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def _init_collectives(rank: int, world: int, local_rank: int, want_rccl: bool, why_not: str = ""):
    """Set up the process group of an N-rank run and say what carries its collectives.  RCCL (backend "nccl") first, checked with one
    all-reduce whose answer is known; if any rank cannot set it up or gets a wrong sum, EVERY rank tears it down and the same processes go on
    over gloo with host tensors (the --ranks-share-device code path) — no re-exec, no restart of a process that has touched its GPU — so a node
    on which RCCL does not come up still ends with a line, flagged.  The ranks agree on the verdict through the rendezvous store, not through
    the group under test.  Returns (device the collectives' tensors live on, what the line's "collectives" says).
    JSP_BENCH_FORCE_RCCL_FAILURE=all (tests) makes the RCCL attempt fail on every rank; =<k> only on rank k."""
    import datetime
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    store, _, _ = next(iter(dist.rendezvous("env://", rank=rank, world_size=world)))
    store.set_timeout(datetime.timedelta(seconds=600))
    if not want_rccl:
        dist.init_process_group("gloo", store=dist.PrefixStore("gloo", store), rank=rank, world_size=world)
        return "cpu", f"gloo ({why_not})"
    reason = None
    try:
        forced = os.environ.get("JSP_BENCH_FORCE_RCCL_FAILURE")
        if forced is not None and (forced == "all" or forced == str(rank)):
            raise RuntimeError("forced by JSP_BENCH_FORCE_RCCL_FAILURE")
        dist.init_process_group("nccl", store=dist.PrefixStore("rccl", store), rank=rank, world_size=world,
                                device_id=torch.device("cuda", local_rank), timeout=datetime.timedelta(seconds=120))
        probe = torch.full((2,), float(rank + 1), dtype=torch.float64, device=f"cuda:{local_rank}")
        dist.all_reduce(probe)
        torch.cuda.synchronize()
        want = world * (world + 1) / 2.0
        if probe.tolist() != [want, want]:
            raise RuntimeError(f"all-reduce over RCCL returned {probe.tolist()}, not {want}")
    except Exception as e:                                    # noqa: BLE001 — whatever RCCL / torch raise here, the run goes on over gloo
        reason = f"{type(e).__name__}: {str(e).splitlines()[0][:160] if str(e) else ''}"
    store.set(f"rccl_verdict_{rank}", reason or "ok")
    verdicts = [store.get(f"rccl_verdict_{r}").decode() for r in range(world)]
    bad = [(r, v) for r, v in enumerate(verdicts) if v != "ok"]
    if not bad:
        return "cuda", "rccl"
    if dist.is_initialized():
        try:
            dist.destroy_process_group()
        except Exception:                                     # noqa: BLE001
            pass
    dist.init_process_group("gloo", store=dist.PrefixStore("gloo_after_rccl", store), rank=rank, world_size=world)
    r0, v0 = bad[0]
    return "cpu", f"gloo (rccl failed on {len(bad)} of {world} ranks; rank {r0}: {v0})"


class Job:
    """What every leg needs to know about the run: ranks, the stream everything is queued on, the bracket."""

    def __init__(self, args, rank, local_rank, world, distributed, stream, coll_device="cuda"):
        self.args, self.rank, self.local_rank, self.world, self.distributed, self.stream = args, rank, local_rank, world, distributed, stream
        self.coll_device = coll_device        # "cuda": RCCL; "cpu": gloo (--ranks-share-device: RCCL does not take two ranks on one device)

    def barrier(self):
        import torch
        if self.distributed:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()


def resident_leg(job: Job, name: str, nfr_override=None, nclips=None):
    """Workload `name` with its inputs resident in HBM: W untimed steps, EXACTLY K timed steps between barrier +
    synchronize on both sides, HIP events on the launch stream around the same region, then every frame the timed kernels
    left in HBM against the oracle's digests.  Returns (result dict, clips); raises SystemExit on a mismatch (any rank)."""
    import torch
    import torch.distributed as dist
    from jsplayer_amd import workloads as wl
    from jsplayer_amd.sharding import gather_per_rank, reduce_counters
    args, rank = job.args, job.rank
    spec = dict(wl.WORKLOADS[name])
    t_gen = time.perf_counter()
    clips = wl.build_clips(name, rank, nfr_override)
    if nclips:   # experiment knob: the first n clips of the workload only (digests still match: clips are independent)
        clips = clips[:max(1, nclips)]
    t_gen = time.perf_counter() - t_gen

    t_stage = time.perf_counter()
    options = dict(kv.split("=", 1) for kv in os.environ.get("JSP_BENCH_OPTIONS", "").split(",") if "=" in kv)    # experiment knob: codec options, "key=value,key=value"
    work = wl.StagedWorkload(name, clips, device=job.local_rank, hip_stream=job.stream.cuda_stream, options=options or None)   # host stage + H2D: outside the timed region
    t_stage = time.perf_counter() - t_stage
    infos, nfr, step = work.infos, work.frames_per_step, work.step
    for _ in range(args.warmup):
        step()
    job.barrier()                             # all ranks ready, device idle
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record(job.stream)
    for _ in range(args.steps):
        step()
    ev1.record(job.stream)
    torch.cuda.synchronize()                  # this rank's K steps are done
    elapsed = time.perf_counter() - t0
    job.barrier()                             # closing bracket; the job's time is the max over ranks (below)
    gpu_ms = ev0.elapsed_time(ev1)            # HIP events on the launch stream, whole timed region
    # ---- what a launch of the timed kind writes, against the oracle's digests ---------------------------------
    # Warm-up already left correct frames in HBM, so hashing them would pass even if the timed replays had done nothing.  Every
    # destination frame is therefore overwritten with 0xEE bytes first, ONE more (untimed) step of exactly the launches that were
    # timed runs, and what THAT leaves is hashed: a frame the replay does not fully write cannot match.
    verified, bad = False, []
    if not args.no_verify and nfr_override is None:
        gold = wl.golden_digests(name, rank)
        if gold is None:
            verified = "no golden digests recorded for this workload / rank"
        else:
            work.scrub()
            step()
            work.sync()
            bad = work.mismatches(gold)
            verified = not bad
    # 2 = every frame matches, 1 = not checked, 0 = mismatch; the job's verdict is the minimum over ranks
    ok = torch.tensor([0 if bad else (2 if verified is True else 1)], dtype=torch.int64, device=job.coll_device)
    if job.distributed:
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    if int(ok[0]) == 0:
        if bad:
            print(f"bench.py rank {rank}: {name}: {len(bad)} frames differ from the oracle's digests (first: clip/frame {bad[0]})", file=sys.stderr)
        if job.distributed:
            dist.destroy_process_group()
        raise SystemExit(f"bench.py: {name}: decoded frames differ from the oracle: no value printed")
    # slow paths the timed launches may have taken instead of the kernels named below (results never depend on them; the
    # number would): the fused kernel's look-back time-out re-runs a batch through the descriptor kernels
    fallbacks = work.lookback_fallbacks()
    fb = torch.tensor([fallbacks], dtype=torch.int64, device=job.coll_device)
    if job.distributed:
        dist.all_reduce(fb, op=dist.ReduceOp.SUM)
    fallbacks = int(fb[0])
    if fallbacks:
        print(f"[bench] {name}: {fallbacks} look-back fall-backs inside the run: the step time includes descriptor-path re-runs", file=sys.stderr)

    total_frames, total_pixels, elapsed = reduce_counters(nfr * args.steps, nfr * args.steps * wl.W * wl.H, elapsed, device=job.coll_device)
    per_rank = gather_per_rank(nfr * args.steps, device=job.coll_device)

    launches = sum(i["kernel_launches"] for i in infos)
    step_us = gpu_ms * 1e3 / args.steps                       # GPU time of one step, HIP events
    alg = sum(i["algorithmic_bytes"] for i in infos)          # SURVEY.md 8(d) formula, per step
    moved = sum(i["moved_bytes"] for i in infos)              # what the launch plan has to move at the least
    counted = min(alg, moved)     # reads the plan provably avoids (previous frame kept in registers) are not credited
    achieved = counted / (step_us * 1e-6) / 1e9               # GB/s
    kernels = work.kernels()
    traffic, traffic_source = None, None
    tpath = os.path.join(ROOT, "profiles", "traffic_by_workload.json")
    if os.path.exists(tpath):    # PMC passes kept under profiles/ (tools/pmc_traffic.sh): used only when they describe these kernels
        try:
            ent = json.load(open(tpath))["per_step"].get(name)
            if ent and ent["kernels"] == kernels and ent["frames_per_step"] == nfr:
                # ... and only while the kernels' SOURCES are the ones the passes were taken from (round 6: until then a name match was enough,
                # and four entries outlived changes to the kernel they describe)
                if ent.get("kernel_sources") == wl.kernel_source_digest(kernels):
                    traffic, traffic_source = ent["hbm_bytes"], ent["source"]
                else:
                    traffic_source = f"stale: the kernel sources changed since {ent['source']} was taken"
        except Exception:
            pass
    write_share = sum(i["pixels"] for i in infos) * 4 / max(counted, 1)      # the frames written, of all bytes counted
    res = {
        "workload": name,
        "spec": spec,
        "value": round(total_pixels / elapsed / 1e6, 1),
        "ms_per_step": round(elapsed * 1e3 / args.steps, 4),
        "verified": True if int(ok[0]) == 2 else (verified if isinstance(verified, str) else False),
        "lookback_fallbacks": fallbacks,
        "frames_per_step": nfr,
        "clips_per_step": len(clips),
        "destination_frames": ({"from": "jsp_pool_create (the product's frame pool: one pool per clip, placed by probing candidate allocations, include/jsplayer_amd.h)",
                                "allocations_tried": [p.attempts for p in work.pools], "probe_GBs": [round(p.store_rate) for p in work.pools],
                                "probe_ms": [round(p.probe_ms, 1) for p in work.pools], "held_while_probing_GB": [round(p.held_bytes / 1e9, 2) for p in work.pools],
                                "candidates_GBs": [[round(r) for r in p.tried_rates] for p in work.pools]}
                               if work.pools else {"from": "one torch tensor per frame" if os.environ.get("JSP_BENCH_FRAME_POOL") == "torch" else "one torch allocation, frames back to back"}),
        "inputs": ("raw stream bytes resident in HBM (on-GPU parse every step)" if spec.get("parse") == "gpu" else
                   "host-built descriptor tables (+ MSVideo1 stream bytes) resident in HBM"),
        "input_bytes_per_step": sum(i["stream_bytes"] if spec.get("parse") == "gpu" else i["descriptor_bytes"] + (i["stream_bytes"] if spec["codec"] == "msv1" else 0) for i in infos),
        "write_share": write_share,
        "roofline": {
            "bound": "hbm",
            "kernel": kernels,
            "achieved": round(achieved, 1),
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4),
            "traffic": traffic,
            "traffic_source": traffic_source,
            "algorithmic_bytes_per_step": alg,
            "moved_bytes_per_step": moved,
            "counted": "algorithmic" if counted == alg else "moved (the plan reads the previous frame once per launch, not once per frame)",
            "step_us": round(step_us, 2),
            "launches_per_step": launches,
        },
        "host_stage": {
            "host_ms_per_step_batch": round(sum(i["host_stage_ms"] for i in infos), 3),
            "h2d_ms_per_step_batch": round(sum(i["h2d_ms"] for i in infos), 3),
            "device_parse_ms_at_staging": round(sum(i["device_parse_ms"] for i in infos), 3),
            "generate_s": round(t_gen, 1),
            "stage_s": round(t_stage, 1),
            "note": "host stage + upload of one step's batches; outside the timed region (inside e2e)",
        },
        "total_frames": total_frames,
        "per_rank_frames": per_rank,
    }
    work.close()
    return res, clips


def e2e_leg(job: Job, name: str, clips, one_stream=True, batch_calls=True, seconds=None):
    """End to end, from the bytes of an AVI file in pinned host memory: AVI walk -> host stage -> H2D -> kernels, one frame per
    call through the asynchronous entry points (the host stage of frame n+1 overlaps the uploads and kernels of frame n), by
    examples/jsp_play — C++ over the C ABI only — for one stream and for one stream per host thread (independent codec
    instances, the way streams shard).  Every stream of the many-stream run plays a file of its own, and all of
    them play for the same length of time (`--seconds`: a stream starts its file over until the time is up, then stops where it
    is), so the rate is one of streams that really run side by side."""
    import subprocess
    import tempfile
    import torch
    import torch.distributed as dist
    from jsplayer_amd import avi
    from jsplayer_amd import workloads as wl
    args, rank, W, H = job.args, job.rank, wl.W, wl.H
    spec = wl.WORKLOADS[name]
    inter = spec.get("mode") == "inter"
    exe = os.path.join(ROOT, "examples", "jsp_play")
    threads = max(1, min(16, (os.cpu_count() or 1) // max(1, args.gpus)))
    ncap = min(len(clips[0].frames), 256 if spec["codec"] == "msv1" else (64 if not inter else 150))   # a bounded sample of the first clip
    # one file per stream: stream 0 plays the first `ncap` frames of this rank's first clip, streams 1.. play clips of their
    # own (seeds of ranks 1000 + rank*16 + 1 ...: independent inputs), shorter ones — what matters is that no two streams read
    # the same bytes
    nshort = max(8, ncap // 8) if not inter else ncap
    secs = seconds if seconds else float(os.environ.get("JSP_BENCH_E2E_SECONDS", 0) or (1.5 if spec["codec"] == "msv1" else 3.0))
    prefetch_mb = float(os.environ.get("JSP_BENCH_PREFETCH_MB", "32"))    # jsp_play --prefetch (MSVideo1: the file goes up in ranges of this size, jsp_prefetch; 0: a copy / a bus read per frame)

    def avi_of(frames, keys, palette):
        return avi.write_avi(W, H, frames, fourcc=b"SCPR" if spec["codec"] == "sp" else b"CRAM",
                             bpp=24 if spec["codec"] == "sp" else spec["bits"], palette=palette, key_flags=keys)
    blobs = [avi_of(clips[0].frames[:ncap], clips[0].keys[:ncap], clips[0].palette)]
    for s_ in range(1, threads):
        c = wl.build_clips(name, 1000 + rank * 16 + s_, nshort)[0]
        blobs.append(avi_of(c.frames, c.keys, c.palette))
    files = []
    try:
        for blob in blobs:
            tf = tempfile.NamedTemporaryFile(suffix=".avi", dir=os.environ.get("TMPDIR", "/tmp"))
            tf.write(blob)
            tf.flush()
            files.append(tf)

        def run(streams):
            names = ",".join(f.name for f in files[:streams])
            res = subprocess.run([exe, names, "--pipelined", "--quiet", "--depth", "8", "--streams", str(streams), "--seconds", str(secs),
                                  "--device", str(job.local_rank), "--prefetch", str(prefetch_mb)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
            if res.returncode != 0:
                raise SystemExit("examples/jsp_play failed: " + res.stderr.decode()[-500:])
            return json.loads(res.stdout.decode().strip().splitlines()[-1])
        one = None
        if one_stream:
            job.barrier()                             # all ranks play at the same time
            one = run(1)
        job.barrier()
        many = run(threads)
    finally:
        for tf in files:
            tf.close()
    batch_api = None
    if batch_calls and not inter and "inter" not in spec and args.gpus == 1:
        # the same sample through the batch calls, from HOST bytes, wall clock to frames in HBM: jsp_stage_batch (ScreenPressor:
        # the host entropy stage takes the batch's groups of pictures side by side; MSVideo1: the bytes are gathered on several
        # threads — or uploaded from where they are when the caller keeps them in pinned memory — and parsed on the GPU) +
        # jsp_staged_decode; one stream, but not one frame per call.  First call (buffers allocated) and re-staging.
        fr = clips[0].frames[:ncap]
        codec = wl.make_codec(name, clips[0].palette, device=job.local_rank)
        dsts = [torch.empty(W * H, dtype=torch.int32, device=f"cuda:{job.local_rank}") for _ in fr]
        torch.cuda.synchronize()
        st, times = None, []
        for _ in range(3):
            t0 = time.perf_counter()
            st = codec.stage_batch(fr, dsts, is_key=clips[0].keys[:ncap], reuse=st)
            st.decode()
            codec.sync()
            times.append(time.perf_counter() - t0)
        batch_api = {"value": round(len(fr) * W * H / min(times[1:]) / 1e6, 1), "unit": "Mpixels/s", "frames": len(fr),
                     "first_call": round(len(fr) * W * H / times[0] / 1e6, 1),
                     "host_stage_ms": round(st.info()["host_stage_ms"], 1),
                     "note": "jsp_stage_batch + jsp_staged_decode of the same frames from host bytes (pageable memory), one stream, wall "
                             "clock; value = re-staging into the same batch object, first_call = including its buffer allocations"}
        st.close()
        codec.StopAndClean()
        del dsts
    # job-wide: the ranks played at the same time, each on its own GPU; rates add up
    tot = torch.tensor([one["mpixels_per_s"] if one else 0.0, many["mpixels_per_s"], one["uploaded_bytes_per_s"] if one else 0.0, many["uploaded_bytes_per_s"]],
                       dtype=torch.float64, device=job.coll_device)
    if job.distributed:
        dist.all_reduce(tot)
    all_threads = {"value": round(float(tot[1]), 1), "unit": "Mpixels/s", "streams": threads * args.gpus, "frames": many["frames"],
                   "seconds": round(many["seconds"], 3), "uploaded_bytes_per_s": round(float(tot[3])),
                   "note": f"{threads} independent streams per GPU (host threads, a codec instance and a FILE OF ITS OWN each: stream 0 the "
                           f"{ncap}-frame sample, the others {nshort}-frame clips of other seeds), every stream playing its file over and over for the same {secs} s"}
    # what the streams' decoders counted on this rank (jsp_counter, since their creation: the untimed pass included): frames the GPU could not settle
    # alone and the host re-ran (a time-out among them would show here), frames that shared a launch, frames found in a prefetched range
    for k in ("async_reruns", "paired_frames", "prefetched_frames"):
        if k in many:
            all_threads[k] = many[k]
    if not one:
        return {"all_threads": all_threads}
    # what the bus itself delivers on this box: pinned host-to-device copies, nothing else queued (jsp_measure_h2d)
    h2d = None
    try:
        import ctypes as C
        from jsplayer_amd import _native as N
        table, best = {}, 0.0
        for mb in (0.5, 4, 64):
            for ns in (1, 2, 4):
                rate = C.c_double(0.0)
                nbytes = int(mb * (1 << 20))
                if N.lib().jsp_measure_h2d(job.local_rank, nbytes, ns, max(2, int((256 << 20) / nbytes / ns)), C.byref(rate)) == 0:
                    table[f"{mb}MB_x{ns}"] = round(rate.value, 1)
                    best = max(best, rate.value)
        if best > 0:
            h2d = {"value": round(best, 1), "unit": "GB/s", "by_copy_size_and_streams": table,
                   "note": "pinned hipMemcpyAsync host-to-device, 0.5 / 4 / 64 MB per copy on 1 / 2 / 4 streams side by side, wall clock, best of three passes (jsp_measure_h2d)"}
    except Exception as e:
        print(f"[bench] bus ceiling not measured: {e}", file=sys.stderr)
    e2e = {"value": round(float(tot[0]), 1), "unit": "Mpixels/s", "streams": args.gpus, "frames": one["frames"],
           "ms_per_frame": round(one["seconds"] * 1e3 / one["frames"], 4),
           "uploaded_bytes_per_s": round(float(tot[2])),
           "all_threads": all_threads,
           "includes": "AVI bytes in pinned host memory -> chunk walk + host stage + H2D + kernels, one frame per call "
                       "(jsp_decompress_*_async / jsp_wait, 8 frames in flight per stream), examples/jsp_play over the C ABI; "
                       "one untimed pass over the file first (codec and buffers set up), all streams start the timed passes together; "
                       "uploaded_bytes_per_s = compressed bytes handed to the decoders per second (what crosses the bus)"
                       + ("; MSVideo1: the file's bytes go up in ranges of %g MB ahead of the frames (jsp_prefetch, every pass over the file anew) "
                          "instead of a copy per frame" % prefetch_mb if spec["codec"] == "msv1" and prefetch_mb > 0 else "")}
    if spec["codec"] == "msv1":
        e2e["prefetch_MB"] = prefetch_mb
    for k in ("async_reruns", "paired_frames", "prefetched_frames"):
        if k in one:
            e2e[k] = one[k]
    if batch_api:
        e2e["batch_api"] = batch_api
    if h2d:
        e2e["h2d_ceiling_GBs"] = h2d
        buses = 1 if args.ranks_share_device else max(1, args.gpus)     # (ranks that share a device share its bus)
        e2e["uploaded_fraction_of_h2d_ceiling"] = round(float(tot[2]) / buses / 1e9 / h2d["value"], 3)
        e2e["all_threads"]["uploaded_fraction_of_h2d_ceiling"] = round(float(tot[3]) / buses / 1e9 / h2d["value"], 3)
    return e2e


def measured_ceilings(job: Job):
    """What THIS box's memory takes right now, measured in this run: nothing but 16-byte stores, one per lane, workgroups in
    address order (the friendliest shape found, profiles/archive/r03_sp_store_lab.txt) over 2 GiB, HIP events on the bench's stream —
    and the rate on file (tools/hbm_ceiling.hip, profiles/hbm_ceiling.json).  The boxes of the pool differ by up to a fifth
    here, so the fraction of this number travels better than the fraction of 8 TB/s."""
    import torch
    live, recorded = None, None
    try:
        import ctypes as C
        from jsplayer_amd import _native as N
        scratch = torch.empty(1 << 29, dtype=torch.int32, device=f"cuda:{job.local_rank}")
        rate = C.c_double(0.0)
        if N.lib().jsp_measure_fill(C.c_void_p(scratch.data_ptr()), C.c_size_t(scratch.numel() * 4), 10, C.byref(rate), C.c_void_p(job.stream.cuda_stream)) == 0:
            live = {"value": round(rate.value, 1), "unit": "GB/s", "kind": "fill", "shape": "16-byte stores, one per lane, 256-lane workgroups in address order, 2 GiB",
                    "source": "measured in this run (jsp_measure_fill: 10 launches, HIP events, best of 3)"}
        del scratch
    except Exception as e:   # (the number is a courtesy: the bench line does not depend on it)
        print(f"[bench] live fill ceiling not measured: {e}", file=sys.stderr)
    cpath = os.path.join(ROOT, "profiles", "hbm_ceiling.json")
    if os.path.exists(cpath):
        try:
            recorded = json.load(open(cpath))
        except Exception:
            recorded = None
    return live, recorded


def _with_ceilings(roofline, write_share, live, recorded):
    """reference_fill / recorded_fill for one workload's roofline object.  A yardstick, NOT a ceiling: a plain one-store-per-lane fill of
    this box's memory, measured in this run — kernels whose stores come in friendlier bursts (all-solid MSVideo1 frames) have been seen
    3 % above it, so it is printed as a rate and no fraction of it is formed (round 6).  The roofline fraction is `frac`, of the 8 TB/s peak."""
    rec = None
    if recorded:
        try:
            kind = "fill" if write_share >= 0.8 else "copy"
            rec = {"value": recorded[kind + "_GBs"], "unit": "GB/s", "kind": kind, "shape": recorded[kind + "_shape"], "source": recorded["source"]}
        except Exception:
            rec = None
    ceiling = live or rec
    out = dict(roofline)
    out["reference_fill"] = ceiling
    out["recorded_fill"] = rec
    return out


def main():
    from jsplayer_amd import workloads as wl
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default=wl.DEFAULT, choices=sorted(wl.WORKLOADS))
    ap.add_argument("--also", default=None, help="comma-separated further workloads for the `also` block (default: the ScreenPressor "
                                                  "configurations, when --workload is the default one)")
    ap.add_argument("--no-also", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-e2e", action="store_true")
    ap.add_argument("--no-verify", action="store_true", help="experiments only: the JSON line then says verified: false")
    ap.add_argument("--ranks-share-device", action="store_true",
                    help="rehearsal of an N-GPU run on ONE GPU: every rank decodes its own streams (seeds, digests, legs: all per rank, as "
                         "on N GPUs) on device 0, the process group is gloo and the collectives run on host tensors (RCCL refuses two ranks "
                         "on one device); the line says ranks_share_device and n_gpus 1")
    ap.add_argument("--dry-run", action="store_true",
                    help="no GPU: the launcher, the process group (gloo) and the counter collectives only; value is null")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(_spawn_ranks(args.gpus))
    also = [] if args.no_also else ([w for w in args.also.split(",") if w] if args.also is not None else
                                    (list(ALSO_DEFAULT) if args.workload == wl.DEFAULT else []))
    for w in also:
        if w not in wl.WORKLOADS:
            raise SystemExit(f"--also: no workload named {w}")

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run, or let bench.py do it (no WORLD_SIZE)")
    # under torch.distributed.run the process group is set up whatever N is (N = 1 included: same code path)
    distributed = world > 1 or ("RANK" in os.environ and "MASTER_PORT" in os.environ)
    from jsplayer_amd.sharding import gather_per_rank, reduce_counters
    if args.dry_run:
        # the multi-rank plumbing without a GPU: every rank "owns" its stream's frames, nothing is decoded
        collectives = "none (one process)"
        if distributed:
            # (without a GPU the RCCL attempt cannot succeed: with --gpus N the dry run is a rehearsal of the fall-back itself, unless told to skip it)
            _, collectives = _init_collectives(rank, world, local_rank, want_rccl=not args.ranks_share_device, why_not="ranks share a device")
            dist.barrier()

        def counters(name):
            spec = wl.WORKLOADS[name]
            nfr = (spec["frames"] - (1 if spec.get("mode") == "inter" else 0)) * spec.get("clips", 1)
            tf, tp, _ = reduce_counters(nfr * args.steps, nfr * args.steps * wl.W * wl.H, 0.0)
            return {"workload": name, "total_frames": tf, "total_pixels": tp, "per_rank_frames": gather_per_rank(nfr * args.steps)}
        head = counters(args.workload)
        rest = [counters(w) for w in also]
        if rank == 0:
            print(json.dumps({"metric": METRIC, "value": None, "unit": "Mpixels/s", "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup,
                              "dry_run": True, "collectives": collectives, "config": {"workload": args.workload, "streams": args.gpus}, "total_frames": head["total_frames"],
                              "total_pixels": head["total_pixels"], "per_rank_frames": head["per_rank_frames"], "also": rest}), flush=True)
        if distributed:
            dist.barrier()
            dist.destroy_process_group()
        return
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path is the product, there is no CPU fallback")
    share = args.ranks_share_device
    if share:
        local_rank = 0                        # every rank on the one device there is
    torch.cuda.set_device(local_rank)
    coll_device, collectives = "cuda", "none (one process)"
    if distributed:
        coll_device, collectives = _init_collectives(rank, world, local_rank, want_rccl=not share, why_not="ranks share a device")

    # a dedicated (non-null) stream: kernels and the timing events are queued on the same one
    stream = torch.cuda.Stream()
    torch.cuda.set_stream(stream)
    job = Job(args, rank, local_rank, world, distributed, stream, coll_device=coll_device)
    t_all = time.perf_counter()

    name = args.workload
    nfr_override = int(os.environ["JSP_BENCH_FRAMES"]) if os.environ.get("JSP_BENCH_FRAMES") else None   # experiment knob
    nclips = int(os.environ["JSP_BENCH_CLIPS"]) if os.environ.get("JSP_BENCH_CLIPS") else None          # experiment knob
    head, clips = resident_leg(job, name, nfr_override, nclips)
    spec = head["spec"]
    e2e = None if args.no_e2e else e2e_leg(job, name, clips)
    live, recorded = measured_ceilings(job) if rank == 0 else (None, None)

    out = None
    if rank == 0:
        out = {
            "metric": METRIC,
            "value": head["value"],
            "unit": "Mpixels/s",
            "n_gpus": 1 if share else args.gpus,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": head["ms_per_step"],
            "higher_is_better": True,
            "scaling": "weak",
            "collectives": collectives,       # what carried the counter reduce: "rccl", or gloo and why
            "vs_baseline": None,
            "dtype": "u8",
            "data": "synthetic",
            "verified": head["verified"],
            "verified_how": None if args.no_verify else
                            "after the timed region every destination frame is overwritten with 0xEE bytes, one more untimed step of the same "
                            "launches runs, and every frame it leaves in HBM is hashed against the CPU oracle's digest (tests/golden/bench_digests.json)",
            "lookback_fallbacks": head["lookback_fallbacks"],
            "config": {
                "workload": name,
                "codec": f"ScreenPressor v{spec['version']}" if spec["codec"] == "sp" else f"MSVideo1_{spec['bits']}bit",
                "frame": f"{wl.W}x{wl.H}",
                "frames_per_step": head["frames_per_step"],
                "clips_per_step": head["clips_per_step"],
                "destination_frames": head["destination_frames"],
                "device": _device_identity(local_rank),
                "streams": args.gpus,
                "sharding": ("REHEARSAL: %d ranks, one independent stream each, all on GPU 0 (gloo, host-tensor collectives)" % args.gpus) if share else
                            "one independent AVI stream per GPU, no data-path collective",
                "inputs": head["inputs"],
                "input_bytes_per_step": head["input_bytes_per_step"],
            },
            "roofline": _with_ceilings(head["roofline"], head["write_share"], live, recorded),
            "host_stage": head["host_stage"],
            "e2e": e2e,
            "total_frames": head["total_frames"],
            "per_rank_frames": head["per_rank_frames"],
        }
        if args.gpus == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(spec, clips[0])
            if e2e:
                out["e2e"]["vs_cpu_one_thread"] = round(e2e["value"] / out["cpu_baseline"]["value"], 2)
                out["e2e"]["vs_cpu_all_cores"] = round(e2e["value"] / out["cpu_baseline"]["all_cores"]["value"], 2)
                out["e2e"]["all_threads"]["vs_cpu_all_cores"] = round(e2e["all_threads"]["value"] / out["cpu_baseline"]["all_cores"]["value"], 2)
    del clips

    # ---- the other configurations the metric names, each to the same bar (same K and W, same bracket, digests, roofline) ----
    entries = []
    for w in also:
        leg, wclips = resident_leg(job, w)
        wspec = leg["spec"]
        many = None if args.no_e2e else e2e_leg(job, w, wclips, one_stream=False, batch_calls=False)
        if rank == 0:
            ent = {
                "workload": w,
                "codec": f"ScreenPressor v{wspec['version']}" if wspec["codec"] == "sp" else f"MSVideo1_{wspec['bits']}bit",
                "value": leg["value"], "unit": "Mpixels/s", "ms_per_step": leg["ms_per_step"], "verified": leg["verified"],
                "frames_per_step": leg["frames_per_step"], "clips_per_step": leg["clips_per_step"],
                "inputs": leg["inputs"], "input_bytes_per_step": leg["input_bytes_per_step"],
                "destination_frames": {k: v for k, v in leg["destination_frames"].items() if k != "from"},
                "roofline": _with_ceilings(leg["roofline"], leg["write_share"], live, recorded),
                "host_stage": {k: v for k, v in leg["host_stage"].items() if k != "note"},
                "total_frames": leg["total_frames"], "per_rank_frames": leg["per_rank_frames"],
            }
            ent["roofline"].pop("recorded_fill", None)
            ent["roofline"]["reference_fill"] = ent["roofline"]["reference_fill"]["value"] if ent["roofline"]["reference_fill"] else None
            if many:
                ent["e2e"] = many
            if args.gpus == 1 and not args.no_cpu_baseline:
                base = cpu_baseline(wspec, wclips[0], budget_s=4.0)
                ent["cpu_baseline"] = {"value": base["value"], "unit": "Mpixels/s", "cores": 1, "kind": "port", "sample": base["sample"],
                                       "all_cores": base["all_cores"]}
                if many:
                    ent["e2e"]["all_threads"]["vs_cpu_all_cores"] = round(many["all_threads"]["value"] / base["all_cores"]["value"], 2)
            entries.append(ent)
        del wclips
    if rank == 0:
        if share:
            out["ranks_share_device"] = True
            out["ranks"] = args.gpus
        if also:
            out["also"] = entries
        out["bench_wall_s"] = round(time.perf_counter() - t_all, 1)
        print(json.dumps(out), flush=True)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
