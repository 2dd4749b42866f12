#!/usr/bin/env python3
"""bench.py — VP9 block-reconstruction throughput on MI355X through the PRODUCT path.

A "step" is one pass of the hot path (inter prediction -> inverse transform + add -> wave-ordered intra
prediction || loop filter) over one 2560x1440 8-bit 4:2:0 frame with a VP9-shaped block structure
(tests/blockgen.py: all 13 block sizes, sub-8x8 blocks with four motion vectors, compound prediction, 4 tile
columns, 8 % intra blocks), packed by the product's C packer (vp9hip_pack_frame) and run by the product's
frame driver (vp9hip_decoder_run) — the code behind wrap_cuda_inter_prediction / wrap_cuda_intra_prediction.
BASELINE.json's configs[1] (Bravia.1440.ivf) is not in the reference snapshot; S-1440 stands in.

  value            frames/s with the frames' work lists + coefficients and the references resident in HBM
                   (vp9hip_decoder_begin_frame done before the timed region; --frames distinct frames cycled
                   through the decoder's ring of list sets)
  pack_upload_run  secondary: every step also packs the frame (C packer, page-locked arrays) and uploads
                   lists + coefficients over PCIe from a page-locked ring while the previous frame's
                   kernels run
  stream           secondary: the reference's own vpxdec linked against libvp9hip_shim.so decoding the
                   synthesized S-1440 IVF end to end (CPU entropy decode included), beside the same vpxdec with
                   the CPU wrap_cuda_* bodies; per-frame MD5s checked
  cpu_baseline     the same frames through the REFERENCE's own C functions (oracle/_ref, ref_recon_frame)
  decode_fps       top level, beside `value`: a real bitstream (S-1440) decoded END TO END by cuda-vp9_amd/vp9hip_dec (own
                   front-end on the CPU + this path on the GPU, every shown frame fetched to the host), and
                   decode_fps_vpxdec the same through the reference's vpxdec on the shim — `value` is the GPU-side ceiling
                   of the hot path with everything resident, these are what a caller gets

`python bench.py --gpus N` starts N ranks itself (one process per GPU, torch.distributed over RCCL, no
data-path collective: independent streams, one per rank); under `torch.distributed.run` (RANK set) it is a
rank.  Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes
import hashlib
import json
import os
import re
import subprocess
import sys
import time

# (before anything initialises the HIP runtime: see cuda-vp9_amd/__init__.py)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
KERNEL_SOURCES = {"convolve": ["inter_kernels.hip"], "idct_add": ["txfm_kernels.hip", "txfm_device.h"],
                  "intra": ["intra_kernels.hip", "txfm_device.h"], "loop_filter": ["lf_kernels.hip"],
                  "walk_lf": ["lf_kernels.hip", "intra_kernels.hip", "txfm_device.h"]}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400,
                    help="timed steps; raised (and reported as run) when they would take less than --min-seconds")
    ap.add_argument("--min-seconds", type=float, default=0.2, help="shortest timed region")
    ap.add_argument("--no-pmc", action="store_true", help="do not re-measure roofline.traffic with rocprofv3 --pmc child runs")
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--width", type=int, default=2560)
    ap.add_argument("--height", type=int, default=1440)
    ap.add_argument("--bit-depth", type=int, default=8)
    ap.add_argument("--frames", type=int, default=4, help="distinct resident frames cycled through (<= 4)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    ap.add_argument("--no-phase-timers", action="store_true")
    ap.add_argument("--no-stream", action="store_true", help="skip the vpxdec end-to-end leg")
    ap.add_argument("--streams", type=int, default=8,
                    help="extra leg: this many independent decoders in flight on one GPU (0 = skip); reported "
                         "under multi_stream, never as `value`")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (gloo with --dry-run on CPU)")
    ap.add_argument("--dry-run", action="store_true",
                    help="host side only (launcher, stream sharding, C packer, stats reduce): no GPU is touched")
    ap.add_argument("--ranks-share-gpu", action="store_true",
                    help="rehearsal of --gpus N on a box with ONE GPU: every rank uses GPU 0, the stats reduce runs on gloo "
                         "(RCCL refuses two ranks on one device); the line says so under config")
    ap.add_argument("--cpu-worker", type=float, default=0.0, help=argparse.SUPPRESS)
    return ap.parse_args()


# ---------------------------------------------------------------------------------------------------
def launch_ranks(args):
    """Parent of `--gpus N`: one child per GPU.  Nothing here initialises a GPU (device_count does not)."""
    import socket
    if not args.dry_run:
        import torch
        have = torch.cuda.device_count()
        if have < args.gpus and not (args.ranks_share_gpu and have >= 1):
            print(json.dumps({"error": f"--gpus {args.gpus} asked for, {have} GPU(s) visible", "n_gpus": args.gpus}))
            return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    for p in procs:
        rc = max(rc, abs(p.wait()))
    return rc


# ---------------------------------------------------------------------------------------------------
def make_frames(hip, W, H, bd, stream, n):
    """n frames of synthetic stream `stream`: decoded blocks, coefficients (reference layout), references."""
    import numpy as np
    import blockgen
    import workload
    import refframe
    rng = np.random.default_rng(1440 + 1000 * stream)
    dt = np.uint16 if bd > 8 else np.uint8
    dims, _ = refframe.plane_dims(W, H)
    refs = [[np.ascontiguousarray(workload.smooth_noise(rng, d[1], d[0], bd, sigma=1.5 + k).astype(dt)) for d in dims]
            for k in range(3)]
    frames = []
    for _ in range(n):
        blocks = blockgen.gen_blocks(rng, W, H, hip.BLOCK_DTYPE, intra_frac=0.08, skip_frac=0.35)
        coef, eob = blockgen.gen_coeffs(rng, blocks, W, H, bd)
        frames.append((blocks, coef, eob))
    return refs, frames


def frame_params(hip, W, H, bd):
    P = hip.FrameParams()
    P.width, P.height, P.ss_x, P.ss_y, P.bit_depth, P.hbd, P.log2_tile_cols, P.build_lf_masks = W, H, 1, 1, bd, int(bd > 8), 2, 1
    for k in range(3):
        P.ref_width[k], P.ref_height[k] = W, H
    return P


def algorithmic_bytes(L, W, H, bd):
    """SURVEY §8(d) per-unit figures over the packed lists of one frame."""
    import numpy as np
    bps = 2 if bd > 8 else 1
    it = L["inter_tasks"]
    px = it["w"].astype(np.int64) * it["h"]
    conv = int((px * (1 + (it["flags"] & 1))).sum() * bps + px.sum() * bps + 32 * len(it))
    tb = L["txb"]
    n2 = (4 << tb["tx_size"].astype(np.int64)) ** 2
    dc = tb["eob"] <= 1
    idct = int((np.where(dc, 1, n2) * 4).sum() + 2 * n2.sum() * bps + 16 * len(tb))
    ia = L["intra_decode_order"]
    bs = 4 << ia["tx_size"].astype(np.int64)
    intra = int((bs * bs * bps).sum() + ((3 * bs + 1) * bps).sum() + 16 * len(ia) + ((ia["eob"] > 0) * bs * bs * 4).sum())
    aw, ah = (W + 7) & ~7, (H + 7) & ~7
    lf = int(2 * (aw * ah * 3 // 2) * bps + 160 * L["sb_rows"] * L["sb_cols"])
    return dict(convolve=conv, idct_add=idct, intra=intra, loop_filter=lf, walk_lf=intra + lf)


def source_hash(files):
    h = hashlib.sha256()
    for f in files:
        with open(os.path.join(ROOT, "cuda-vp9_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def cpu_worker(args):
    """Child of the N-process CPU baseline: no GPU; prints frames and seconds."""
    import __graft_entry__ as g
    import refframe
    hip = g.load_pkg()
    refs, frames = make_frames(hip, args.width, args.height, args.bit_depth, 0, 1)
    rf = refframe.RefFrame(refframe.load_ref(), frames[0][0], args.width, args.height, args.bit_depth, refs,
                           [(args.width, args.height)] * 3, frames[0][1], frames[0][2], tiles=2)
    n, t = 0, 0.0
    while t < args.cpu_worker or n < 2:
        t += rf.run()
        n += 1
    print(json.dumps({"frames": n, "seconds": t}))
    return 0


def run_vpxdec(path, ivf, loops=1, md5=False, timeout=600):
    cmd = [path] + (["--rawvideo", "--md5", "-o", "img-%wx%h-%4.i420"] if md5 else ["--noblit", "--summary", f"--loops={loops}"]) + [ivf]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=timeout)
    out = r.stdout.decode(errors="replace")
    if r.returncode:
        raise RuntimeError(f"{' '.join(cmd)} failed ({r.returncode}): {out[-400:]}")
    if md5:
        return [l for l in out.splitlines() if re.match(r"^[0-9a-f]{32}  img-", l)]
    return [(int(m.group(1)), float(m.group(2))) for m in re.finditer(r"(\d+) decoded frames/\d+ showed frames in \d+ us \(([\d.]+) fps\)", out)]


def run_own_dec(ivf, loops=1, md5=False, timeout=600, device=0, threads=None):
    """cuda-vp9_amd/vp9hip_dec: the decoder built only from this repository (own bitstream front-end + GPU
    reconstruction).  md5: vpxdec's per-frame lines; else [(frames, fps)] per loop, frames fetched to the host."""
    path = os.path.join(ROOT, "cuda-vp9_amd", "vp9hip_dec")
    cmd = [path, f"--device={device}"] + ([f"--threads={threads}"] if threads else []) + \
          (["--md5", "-o", "img-%wx%h-%4.i420"] if md5 else ["--noblit", "--fetch", "--summary", f"--loops={loops}"]) + [ivf]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=timeout)
    out = r.stdout.decode(errors="replace")
    if r.returncode:
        raise RuntimeError(f"{' '.join(cmd)} failed ({r.returncode}): {out[-400:]}")
    if md5:
        return [l for l in out.splitlines() if re.match(r"^[0-9a-f]{32}  img-", l)]
    return [(int(m.group(1)), float(m.group(2))) for m in re.finditer(r"(\d+) decoded frames/\d+ showed frames in \d+ us \(([\d.]+) fps\)", out)]


def real_stream_resident(hip, ivf, replay=(7, 14, 21, 28, 35, 42, 49, 56), reps=150):
    """GPU-side rate on the frames of a real stream: the stream is decoded once (own front-end + GPU), and a few of
    its frames are then replayed with their work lists resident in HBM (ring set re-selected, same references) —
    what the reconstruction sustains on real content when the CPU entropy stage is not in the way."""
    import torch
    import time as _t
    dec = hip.Decoder(0)
    fe = hip.FrontEnd(threads=0, decoder=dec)
    dec.set_timing(True)
    rates, index = [], 0
    try:
        for pkt in hip.ivf_packets(ivf):
            for data in fe.frames_of(pkt):
                fr = fe.parse(data)
                if fr.show_existing:
                    continue
                ring = dec.begin_parsed(fr)
                dec.run_parsed(fr)
                dec.sync()
                if index in replay:
                    for _ in range(10):
                        dec.select_set(ring)
                        dec.run_parsed(fr)
                    dec.sync()
                    dec.set_timing(False)
                    t0 = _t.perf_counter()
                    for _ in range(reps):
                        dec.select_set(ring)
                        dec.run_parsed(fr)
                    dec.sync()
                    rates.append((index, fr.n_blocks, int(fr.filter_level), reps / (_t.perf_counter() - t0)))
                    dec.set_timing(True)
                index += 1
    finally:
        fe.close()
        dec.close()
    if not rates:
        return None
    hm = len(rates) / sum(1.0 / r[3] for r in rates)
    return {"frames_per_s": round(hm, 1), "frames": [{"index": i, "blocks": n, "filter_level": l, "frames_per_s": round(r, 1)} for i, n, l, r in rates],
            "note": "frames of the S-1440 stream as vp9hip_fe parsed them, packed by vp9hip_pack_frame, lists + coefficients resident, "
                    "vp9hip_decoder_run replayed %d times each (harmonic mean)" % reps}


def stream_pair(name, threads=None):
    """One BASELINE-sized stream through cuda-vp9_amd/vp9hip_dec (frames fetched) and the reference's CPU path, MD5 checked."""
    big = os.path.join(ROOT, "tests", "streams_big")
    ivf, gold = os.path.join(big, name + ".ivf"), os.path.join(big, name + ".md5")
    cdec = os.path.join(ROOT, "oracle", "_ref", "vpx", "vpxdec_c")
    if not (os.path.exists(ivf) and os.path.exists(gold) and os.path.exists(os.path.join(ROOT, "cuda-vp9_amd", "vp9hip_dec"))):
        return {"skipped": f"tests/streams_big/{name}.ivf or vp9hip_dec absent"}
    want = [l.rstrip("\n") for l in open(gold) if l.strip()]
    runs = run_own_dec(ivf, loops=5)
    out = {"frames": len(want), "vp9hip_dec_fps": round(sum(f for _, f in runs[1:]) / max(1, len(runs[1:])), 2),
           "md5_match": run_own_dec(ivf, md5=True) == want}
    if os.path.exists(cdec):
        out["vpxdec_c_fps"] = run_vpxdec(cdec, ivf, loops=1)[0][1]
    return out


def measure_traffic(kernel_key):
    """HBM bytes per launch of one kernel family from two rocprofv3 --pmc child runs (FETCH_SIZE and WRITE_SIZE do
    not fit one pass on gfx950; FETCH_SIZE tallies a 128-byte request as 64 bytes: doubled, MI355X_MICROARCH.md
    "HBM") of tools/profile_phase.py on the same frame.  None when rocprofv3 is absent or a pass fails."""
    import csv
    import glob
    import shutil
    import tempfile
    if not shutil.which("rocprofv3"):
        return None, "rocprofv3 not on PATH"
    vals = {}
    with tempfile.TemporaryDirectory(dir="/tmp") as tmp:
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            # the interpreter itself after `--`: rocprofv3's preloaded library has initialised the GPU before the program starts
            cmd = ["rocprofv3", "--pmc", ctr, "--output-format", "csv", "-d", os.path.join(tmp, ctr), "-o", "p", "--",
                   sys.executable, os.path.join(ROOT, "tools", "profile_phase.py"), "--separate", "--steps", "4"]
            try:
                r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=240, cwd="/tmp",
                                   env=dict(os.environ, TMPDIR="/tmp"))
            except (subprocess.TimeoutExpired, OSError) as e:
                return None, f"rocprofv3 --pmc {ctr}: {e}"
            if r.returncode:
                return None, f"rocprofv3 --pmc {ctr} failed ({r.returncode})"
            per = {}
            for f in glob.glob(os.path.join(tmp, ctr, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    if kernel_key in row["Kernel_Name"] and row["Counter_Name"] == ctr:
                        per[row["Dispatch_Id"]] = per.get(row["Dispatch_Id"], 0.0) + float(row["Counter_Value"])
            if not per:
                return None, f"no {kernel_key} dispatch in the {ctr} pass"
            vals[ctr] = sum(per.values()) / len(per)
    return int(vals["FETCH_SIZE"] * 2048 + vals["WRITE_SIZE"] * 1024), "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE child runs of tools/profile_phase.py, this box, this build (FETCH_SIZE x 2 on gfx950)"


def stream_leg():
    """The reference's vpxdec end to end on the synthesized S-1440 IVF: HIP-linked vs CPU bodies."""
    big = os.path.join(ROOT, "tests", "streams_big")
    ivf, gold = os.path.join(big, "S-1440.ivf"), os.path.join(big, "S-1440.md5")
    hipdec, cdec = os.path.join(ROOT, "shim", "build", "vpxdec_hip"), os.path.join(ROOT, "oracle", "_ref", "vpx", "vpxdec_c")
    mtdec = os.path.join(ROOT, "shim", "build", "vpxdec_hip_mt")
    if not all(os.path.exists(p) for p in (ivf, gold, hipdec, cdec)):
        return {"skipped": "tests/streams_big/S-1440.ivf or the vpxdec builds are absent (tests/golden/streams/make_streams.py --big, oracle/build_refvpx.sh)"}
    want = [l.rstrip("\n") for l in open(gold) if l.strip()]
    got = run_vpxdec(hipdec, ivf, md5=True)
    got_mt = run_vpxdec(mtdec, ivf, md5=True) if os.path.exists(mtdec) else None
    mt_runs = run_vpxdec(mtdec, ivf, loops=4) if os.path.exists(mtdec) else None
    hip_runs = run_vpxdec(hipdec, ivf, loops=4)   # first loop pays the HIP start-up; report the warm ones
    c_runs = run_vpxdec(cdec, ivf, loops=1)
    warm = hip_runs[1:] or hip_runs
    own = {}
    if os.path.exists(os.path.join(ROOT, "cuda-vp9_amd", "vp9hip_dec")):
        own_runs = run_own_dec(ivf, loops=5)
        own = {"vp9hip_dec_fps": round(sum(f for _, f in own_runs[1:]) / max(1, len(own_runs[1:])), 2),
               "vp9hip_dec_md5_match": run_own_dec(ivf, md5=True) == want,
               "vp9hip_dec_note": "no libvpx: the repository's own bitstream front-end (vp9hip_fe, one thread per tile column) + GPU "
                                  "reconstruction, parsing thread and GPU thread pipelined, every shown frame fetched to the host"}
    return {"stream": "S-1440 (synthesized 2560x1440 8-bit IVF, %d frames; tests/golden/streams/make_streams.py)" % len(want),
            "md5_frames_equal": sum(a == b for a, b in zip(got, want)), "md5_frames": len(want),
            "md5_match": got == want,
            "vpxdec_hip_fps": round(sum(f for _, f in warm) / len(warm), 2), "vpxdec_hip_fps_first_loop": hip_runs[0][1],
            "vpxdec_hip_mt_fps": round(sum(f for _, f in mt_runs[1:]) / len(mt_runs[1:]), 2) if mt_runs else None,
            "vpxdec_hip_mt_md5_match": (got_mt == want) if got_mt is not None else None,
            "vpxdec_c_fps": c_runs[0][1], **own,
            "note": "dx_time-based fps printed by vpxdec --summary (libvpx/vpxdec.c:358-363): whole decode incl. CPU entropy "
                    "stage; vpxdec_c and vpxdec_hip parse on one thread, vpxdec_hip_mt with one thread per tile column (8 here); "
                    "HIP = patched frame driver (INTEGRATION.md mode C)"}


# ---------------------------------------------------------------------------------------------------
def main():
    args = parse_args()
    # host packer threads: 8 of the 16 cores a GPU's share of the box has (cuda-vp9_amd/csrc/vp9hip_pack.c)
    os.environ.setdefault("VP9HIP_PACK_THREADS", "8")
    if args.cpu_worker > 0:
        return cpu_worker(args)
    if args.gpus > 1 and "RANK" not in os.environ:
        return launch_ranks(args)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"WORLD_SIZE {world} != --gpus {args.gpus}")

    import numpy as np
    import torch
    import __graft_entry__ as g
    hip = g.load_pkg()
    import cuda_vp9_amd.batch as batch

    # several ranks on one host: a block of CPUs per rank on its GPU's NUMA node and thread budgets from that block
    # (SURVEY §8e); a single rank keeps the whole host (the CPU baseline's child processes want it)
    placement = None
    if world > 1:
        placement = batch.rank_placement(local_rank, world, batch.host_topology())
        placement["pinned"] = batch.apply_placement(placement)
        os.environ["VP9HIP_PACK_THREADS"] = str(placement["pack_threads"])

    dist = None
    gpu = 0 if args.ranks_share_gpu else local_rank  # device index of this rank
    if args.ranks_share_gpu:
        args.backend = "gloo"
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dry_run or args.ranks_share_gpu:
            if not args.dry_run:
                torch.cuda.set_device(gpu)
            dist.init_process_group(backend=args.backend)
        else:
            torch.cuda.set_device(gpu)
            dist.init_process_group(backend=args.backend, device_id=torch.device("cuda", gpu))
    elif not args.dry_run:
        torch.cuda.set_device(gpu)
    dev = None if (args.dry_run or args.ranks_share_gpu) else torch.device("cuda", gpu)  # where the stats reduce runs

    W, H, bd = args.width, args.height, args.bit_depth
    n_frames = max(1, min(args.frames, 4))
    # independent streams, one per GPU: stream i -> rank i mod world
    my_streams = batch.shard_streams(world, rank, world)
    refs, frames = make_frames(hip, W, H, bd, my_streams[0], n_frames)
    P = frame_params(hip, W, H, bd)
    th = hip.LfThresh()
    hip.lib().vp9hip_lf_frame_init(32, 0, None, None, 0, 0, None, None, None, ctypes.byref(th))
    pk = hip.Packer()
    lists0 = pk.pack(P, frames[0][0], frames[0][2])
    ab = algorithmic_bytes(lists0, W, H, bd)
    tq0 = time.perf_counter()
    for i in range(8):
        pk.pack_only(P, frames[i % n_frames][0], frames[i % n_frames][2])
    t_pack = (time.perf_counter() - tq0) / 8
    pk.close()

    def barrier():
        if dist is not None:
            dist.barrier()
        if not args.dry_run:
            torch.cuda.synchronize()

    if args.dry_run:
        barrier()
        total, _, _ = batch.reduce_stats(dist, float(len(frames)), 0.0, 1.0, device=None)
        if rank == 0:
            print(json.dumps({"metric": "dry run: launcher + stream sharding + C packer + stats reduce, no GPU",
                              "value": None, "n_gpus": world, "frames_packed_all_ranks": total,
                              "streams_of_rank0": my_streams, "inter_tasks_frame0": int(len(lists0["inter_tasks"])),
                              "host_pack_ms_per_frame": round(t_pack * 1e3, 3)}))
        if dist is not None:
            dist.destroy_process_group()
        return 0

    ALL = hip.PHASE_INTER | hip.PHASE_INTRA | hip.PHASE_LF
    dec = hip.Decoder(gpu)
    for k in range(3):
        dec.upload(k, refs[k], W, H, bd)
    dec.alloc_slot(3, W, H, bd)
    # coefficients in page-locked memory (what the shim hands the caller's entropy stage to write into)
    pinned = []
    for (blocks, coef, eob) in frames:
        pc = []
        for c in coef:
            a = dec.host_array(max(1, len(c)), np.int32)
            a[:len(c)] = c
            pc.append(a[:len(c)])
        pinned.append(pc)
    sets = [dec.begin_frame(P, fr[0], fr[2], pc, persistent=True) for fr, pc in zip(frames, pinned)]
    dec.sync()
    dec.set_timing(False)  # the timed legs do not read per-run GPU times

    def step(i):
        dec.select_set(sets[i % n_frames])
        dec.run(ALL, (0, 1, 2), 3, thresh=th)

    # ---- the timed region: K steps on resident lists ------------------------------------------------
    # K = --steps, raised so that the region lasts at least --min-seconds (a 20-step region is 10 ms of GPU time: too
    # short to mean anything); decided from the untimed warm-up, the same K on every rank, reported as `steps`
    tw0 = time.perf_counter()
    for i in range(max(1, args.warmup)):
        step(i)
    dec.sync()
    per_step = (time.perf_counter() - tw0) / max(1, args.warmup)
    steps = max(args.steps, int(args.min_seconds / max(per_step, 1e-6)) + 1)
    if dist is not None:
        k = torch.tensor([steps], dtype=torch.int64, device=dev)
        dist.all_reduce(k, op=dist.ReduceOp.MAX)
        steps = int(k.item())
    barrier()
    t0 = time.perf_counter()
    for i in range(steps):
        step(i)
    barrier()
    elapsed = time.perf_counter() - t0
    dec.sync()  # reports a loop-filter row that gave up waiting, if any

    # ---- secondary: pack + PCIe upload + run per step, pipelined through the ring ---------------------
    n_pipe = max(20, min(steps, 400) // 4)
    for i in range(4):
        dec.begin_frame(P, frames[i % n_frames][0], frames[i % n_frames][2], pinned[i % n_frames], persistent=True)
        dec.run(ALL, (0, 1, 2), 3, thresh=th)
    barrier()
    tp0 = time.perf_counter()
    for i in range(n_pipe):
        dec.begin_frame(P, frames[i % n_frames][0], frames[i % n_frames][2], pinned[i % n_frames], persistent=True)
        dec.run(ALL, (0, 1, 2), 3, thresh=th)
    barrier()
    t_pipe = time.perf_counter() - tp0
    dec.sync()
    sets = [dec.begin_frame(P, fr[0], fr[2], pc, persistent=True) for fr, pc in zip(frames, pinned)]
    dec.sync()

    # ---- per-kernel GPU time: HIP events (the decoder's launch stream) around each phase on its own --
    # walk_lf = the island walk and the loop filter as the product runs them: ONE launch (walk_lf_kernel); intra and
    # loop_filter alone are the two halves run as launches of their own, for reference
    PH = (("convolve", hip.PHASE_INTER_PRED), ("idct_add", hip.PHASE_INTER_RESID), ("walk_lf", hip.PHASE_INTRA | hip.PHASE_LF),
          ("intra", hip.PHASE_INTRA), ("loop_filter", hip.PHASE_LF))
    phase_ms = {}
    if not args.no_phase_timers:
        n_timed = min(steps, 100)
        dec.set_timing(True)
        dec.select_set(sets[0])
        for name, bits in PH:
            tot = 0.0
            for _ in range(n_timed):
                dec.run(bits, (0, 1, 2), 3, thresh=th)
                dec.sync()
                tot += dec.last_run_ms()
            phase_ms[name] = tot / n_timed

    # ---- correctness of what was timed: frame 0 against the reference's own C functions ---------------
    md5_match, cpu_baseline, stream = None, None, None
    extra_streams = {}
    if rank == 0:
        import refframe
        dec.select_set(sets[0])
        dec.run(ALL, (0, 1, 2), 3, thresh=th)
        dec.sync()
        dims, _ = refframe.plane_dims(W, H)
        got = [np.zeros((d[1], d[0]), np.uint16 if bd > 8 else np.uint8) for d in dims]
        dec.download(3, got, W, H, bd)
        rf = refframe.RefFrame(refframe.load_ref(), frames[0][0], W, H, bd, refs, [(W, H)] * 3, frames[0][1], frames[0][2], tiles=2)
        t_ref = rf.run()
        md5_match = refframe.frame_md5(got, W, H) == refframe.frame_md5(rf.planes(), W, H)
        if world == 1 and not args.no_cpu_baseline:
            n, t_cpu = 1, t_ref
            while t_cpu < args.cpu_seconds:
                t_cpu += rf.run()
                n += 1
            cores = min(16, os.cpu_count() or 1)
            kids = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-worker", str(args.cpu_seconds),
                                      "--width", str(W), "--height", str(H), "--bit-depth", str(bd)],
                                     stdout=subprocess.PIPE) for _ in range(cores)]
            outs = [json.loads(k.communicate()[0].decode().strip().splitlines()[-1]) for k in kids]
            cpu_baseline = {"value": round(n / t_cpu, 3), "unit": "frames/s", "cores": 1, "kind": "reference",
                            "sample": f"{n} passes over frame 0 of the GPU run ({W}x{H}, same blocks + coefficients) through the "
                                      f"reference's own C functions (oracle/_ref/libvpxref.so: vp9_build_inter_predictors_sb, "
                                      f"vp9_predict_intra_block, vpx_idct*/vp9_iht*_add_c, vp9_filter_block_plane_*), one thread, {t_cpu:.1f} s",
                            "all_cores": {"value": round(sum(o["frames"] / o["seconds"] for o in outs), 2), "cores": cores,
                                          "sample": f"{cores} single-thread processes side by side, {args.cpu_seconds:.0f} s each"}}
        if world == 1 and not args.no_stream:
            try:
                stream = stream_leg()
            except (RuntimeError, subprocess.TimeoutExpired, OSError) as e:
                stream = {"error": str(e)[:400]}
            ivf_big = os.path.join(ROOT, "tests", "streams_big", "S-1440.ivf")
            if isinstance(stream, dict) and os.path.exists(ivf_big):
                try:
                    stream["resident_replay"] = real_stream_resident(hip, ivf_big)
                except Exception as e:  # a secondary figure: never takes the line down
                    stream["resident_replay"] = {"error": str(e)[:300]}
            # the north star's other size and the headline size at 10 bits, end to end, beside the C path
            for key, nm in (("stream_1080_8", "S-1080-8"), ("stream_1440_10", "S-1440-10")):
                try:
                    extra_streams[key] = stream_pair(nm)
                except (RuntimeError, subprocess.TimeoutExpired, OSError) as e:
                    extra_streams[key] = {"error": str(e)[:300]}

    # ---- extra leg: several independent decoders in flight on one GPU ---------------------------------
    multi = None
    if args.streams > 1:
        barriers = 0  # every rank passes both barriers of this leg, also one that failed on the way
        try:  # a secondary figure: whatever happens in it, the line with `value` is printed
            decs = []
            for s in range(args.streams):
                d = hip.Decoder(gpu)
                for k in range(3):
                    d.upload(k, refs[k], W, H, bd)
                d.alloc_slot(3, W, H, bd)
                d.begin_frame(P, frames[s % n_frames][0], frames[s % n_frames][2], pinned[s % n_frames], persistent=True)
                d.set_timing(False)
                decs.append(d)
            for _ in range(3):
                for d in decs:
                    d.run(ALL, (0, 1, 2), 3, thresh=th)
            for d in decs:
                d.sync()
            barrier()
            barriers = 1
            n_rounds = max(10, min(steps, 400) // 8)
            import threading
            failures = []

            def drive(d):  # one host thread per stream, as one decoder process / thread per stream would (ctypes drops the GIL)
                try:
                    for _ in range(n_rounds):
                        d.run(ALL, (0, 1, 2), 3, thresh=th)
                    d.sync()
                except Exception as e:  # noqa: BLE001
                    failures.append(str(e)[:300])

            thr = [threading.Thread(target=drive, args=(d,)) for d in decs]
            tm0 = time.perf_counter()
            for x in thr:
                x.start()
            for x in thr:
                x.join()
            barrier()
            barriers = 2
            tm = time.perf_counter() - tm0
            multi = {"streams": args.streams, "frames": n_rounds * args.streams, "frames_per_s": round(n_rounds * args.streams / tm, 1),
                     "hw_queues": os.environ.get("GPU_MAX_HW_QUEUES")}
            if failures:
                multi = {"error": failures[0]}
            for d in decs:
                d.close()
        except Exception as e:  # noqa: BLE001
            multi = {"error": str(e)[:300]}
            for _ in range(2 - barriers):
                barrier()

    # ---- extra leg (N = 1): the hot path on the north star's other geometries, same procedure as `value` ----------
    other_geometries = None
    if world == 1 and not args.no_stream and (W, H, bd) == (2560, 1440, 8):
        other_geometries = {}
        for key, (w2, h2, b2) in (("1080p_8bit", (1920, 1080, 8)), ("1440p_10bit", (2560, 1440, 10))):
            try:
                refs2, frames2 = make_frames(hip, w2, h2, b2, 0, 2)
                P2 = frame_params(hip, w2, h2, b2)
                d2 = hip.Decoder(gpu)
                for k in range(3):
                    d2.upload(k, refs2[k], w2, h2, b2)
                d2.alloc_slot(3, w2, h2, b2)
                d2.begin_frame(P2, frames2[0][0], frames2[0][2], frames2[0][1])
                d2.set_timing(False)
                for _ in range(10):
                    d2.run(ALL, (0, 1, 2), 3, thresh=th)
                d2.sync()
                n2, t2 = 0, time.perf_counter()
                while n2 < 200 or time.perf_counter() - t2 < 0.2:
                    d2.run(ALL, (0, 1, 2), 3, thresh=th)
                    n2 += 1
                d2.sync()
                t2 = time.perf_counter() - t2
                import refframe as _rf
                rf2 = _rf.RefFrame(_rf.load_ref(), frames2[0][0], w2, h2, b2, refs2, [(w2, h2)] * 3, frames2[0][1], frames2[0][2], tiles=2)
                rf2.run()
                dims2, _ = _rf.plane_dims(w2, h2)
                got2 = [np.zeros((dd[1], dd[0]), np.uint16 if b2 > 8 else np.uint8) for dd in dims2]
                d2.download(3, got2, w2, h2, b2)
                other_geometries[key] = {"frames_per_s": round(n2 / t2, 1), "steps": n2,
                                         "md5_match": _rf.frame_md5(got2, w2, h2) == _rf.frame_md5(rf2.planes(), w2, h2)}
                d2.close()
            except Exception as e:  # noqa: BLE001
                other_geometries[key] = {"error": str(e)[:300]}

    # ---- extra leg (N > 1): the north star's batch — one real stream per GPU through the stand-alone decoder ------
    streams_per_gpu = None
    ivf_big = os.path.join(ROOT, "tests", "streams_big", "S-1440.ivf")
    have_own = os.path.exists(ivf_big) and os.path.exists(os.path.join(ROOT, "cuda-vp9_amd", "vp9hip_dec"))
    if world > 1 and not args.no_stream and not args.dry_run:
        fps_mine, frames_mine = 0.0, 0
        if have_own:
            try:  # whatever happens here, every rank reaches the collective below
                runs = run_own_dec(ivf_big, loops=5, device=gpu, timeout=300, threads=placement["entropy_threads"] if placement else None)
                warm = runs[1:] or runs
                if warm:
                    fps_mine, frames_mine = sum(f for _, f in warm) / len(warm), warm[0][0]
            except Exception:  # noqa: BLE001
                fps_mine, frames_mine = 0.0, 0
        # sum of the ranks' rates: the streams are independent, every rank decodes its own copy on its own GPU
        tot, _, _ = batch.reduce_stats(dist, fps_mine, 0.0, 1.0, device=dev)
        streams_per_gpu = {"frames_per_s_all_gpus": round(float(tot), 1), "streams": world, "frames_per_stream": frames_mine,
                           "note": "S-1440 (real bitstream) through cuda-vp9_amd/vp9hip_dec on every GPU at once, one stream per GPU, "
                                   "frames fetched to the host; sum of the ranks' warm-loop rates"}

    # stats reduce: total frames (sum) and slowest rank (max) — the only collective in the harness
    frames_total, _, t_max = batch.reduce_stats(dist, steps, 0.0, elapsed, device=dev)
    pipe_total, _, t_pipe_max = batch.reduce_stats(dist, n_pipe, 0.0, t_pipe, device=dev)

    if rank == 0:
        kernels = {}
        for name, _ in PH:
            if name in phase_ms:
                ms = phase_ms[name]
                gbs = ab[name] / (ms * 1e-3) / 1e9 if ms > 0 else None
                kernels[name] = {"ms_per_frame": round(ms, 5), "algorithmic_bytes": ab[name], "GB/s": round(gbs, 1) if gbs else None,
                                 "frac_of_hbm_peak": round(gbs / HBM_PEAK_GBS, 5) if gbs else None}
        roofline = None
        if kernels:
            # the launches of a frame as the product runs it: convolve, idct_add, walk_lf (island walk + loop filter)
            dom = max((k for k in kernels if k in ("convolve", "idct_add", "walk_lf")), key=lambda k: kernels[k]["ms_per_frame"])
            traffic, note = None, "not measured"
            if not args.no_pmc and world == 1 and (W, H, bd) == (2560, 1440, 8):  # (N > 1: the committed passes; no profiler beside other ranks)
                key = {"convolve": "inter_reg_kernel", "idct_add": "idct_add_all_kernel", "walk_lf": "walk_lf_kernel"}[dom]
                try:
                    traffic, note = measure_traffic(key)
                except Exception as e:  # noqa: BLE001 — a secondary figure
                    traffic, note = None, f"PMC child run failed: {str(e)[:200]}"
            if traffic is None:
                # fallback: the committed record of tools/pmc_collect.sh, quoted only while the kernel's source is what was profiled
                try:
                    pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_current.json")))
                    e = pmc.get(dom)
                    if e and (W, H, bd) == (2560, 1440, 8) and e.get("source_sha256") == source_hash(KERNEL_SOURCES[dom]):
                        traffic = int(e["hbm_read_bytes_gfx950_x2"] + e["hbm_write_bytes"])
                        note = f"{note}; profiles/pmc_current.json ({e.get('collected', '')}), kernel source hash matches"
                except (OSError, KeyError, ValueError):
                    pass
            roofline = {"kernel": dom, "bound": "hbm", "achieved": kernels[dom]["GB/s"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": kernels[dom]["frac_of_hbm_peak"], "traffic": traffic, "traffic_source": note}
        blocks0 = frames[0][0]
        out = {
            "metric": "decoded frames/sec (block-reconstruction path: inter+idct+intra+loop filter), 1440p VP9 8-bit",
            "value": round(frames_total / t_max, 2), "unit": "frames/s", "n_gpus": world, "steps": steps,
            "steps_requested": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * t_max / steps, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "u8" if bd == 8 else "u16", "data": "synthetic",
            "config": {"workload": f"S-{H}: synthetic {W}x{H} {bd}-bit 4:2:0 inter frames with a VP9 partition ({len(blocks0)} "
                                   f"blocks, {len(lists0['inter_tasks'])} inter tasks, {len(lists0['txb'])} coded inter tx blocks, "
                                   f"{len(lists0['intra_decode_order'])} intra tx blocks in {lists0['n_waves']} waves, "
                                   f"{lists0['sb_rows']}x{lists0['sb_cols']} SBs) packed by vp9hip_pack_frame, run by "
                                   f"vp9hip_decoder_run; {n_frames} distinct frames resident in HBM, one stream per GPU",
                       "parallelism": f"streams{world}" + (" (rehearsal: all ranks on GPU 0, gloo)" if args.ranks_share_gpu else "")},
            "md5_match_vs_reference_c": md5_match,
            "roofline": roofline, "kernels": kernels, "cpu_baseline": cpu_baseline,
            "pack_upload_run": {"frames_per_s": round(pipe_total / t_pipe_max, 1), "frames": n_pipe,
                                "host_pack_ms_per_frame": round(t_pack * 1e3, 3),
                                "host_pack_threads": int(os.environ["VP9HIP_PACK_THREADS"]),
                                "note": "every step packs the frame on the host and uploads lists + coefficients from "
                                        "page-locked memory (ring of 4 list sets) while the previous frame's kernels run"},
            # a real bitstream end to end (CPU entropy stage + this path, frames fetched), beside `value` — the GPU-side
            # rate of the hot path with everything resident
            "decode_fps": (stream or {}).get("vp9hip_dec_fps"), "decode_fps_vpxdec": (stream or {}).get("vpxdec_hip_mt_fps"),
            "decode_fps_reference_c": (stream or {}).get("vpxdec_c_fps"),
            "stream": stream, **extra_streams, "multi_stream": multi,
            # several decoders in flight on one GPU must never make a filter row give up (tests/test_gpu_multi_decoder.py)
            "multi_stream_ok": (None if multi is None else "error" not in multi),
            "hot_path_other_geometries": other_geometries,
            "stream_per_gpu": streams_per_gpu, "placement": placement,
        }
        print(json.dumps(out))
    dec.close()
    if dist is not None:
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
