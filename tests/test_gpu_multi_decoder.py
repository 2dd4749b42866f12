"""Several decoders on ONE GPU at once.  The fused island walk + loop filter launch (vp9hip_intra_islands_lf) is
built so that a workgroup only ever waits for workgroups in front of it in its own grid (islands wait for nothing;
a filter row waits for the row above and for the islands of superblock rows r, r + 1, all placed before it): whatever
else is in flight on the GPU — other contexts of the process, other processes — every wait ends.  Round 2's form
(filter rows first, spinning on islands dispatched after them) gave up under exactly this load
(gpurun_out/b2.json: "a superblock row gave up waiting").

 * test_decoders_in_one_process: N decoders, one host thread each, dense frames with a VP9 partition (blockgen),
   many frames in flight per decoder; every decoder's last frame bit-equal to the reference's own C functions
   (oracle/_ref through refframe), vp9hip_sync reports no row that gave up.
 * test_first_launch_of_a_context_on_a_busy_gpu: decoders created and destroyed beside two busy ones.
 * test_decoder_processes_side_by_side: N vp9hip_dec processes on the same stream side by side, every loop's MD5
   lines equal to the golden list (what tools/multi_process_check.sh ran by hand in round 2)."""
import ctypes
import os
import subprocess
import threading

import numpy as np
import pytest

import blockgen

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _params(hip, W, H, bd):
    P = hip.FrameParams()
    P.width, P.height, P.ss_x, P.ss_y, P.bit_depth, P.hbd, P.log2_tile_cols, P.build_lf_masks = W, H, 1, 1, bd, int(bd > 8), 2, 1
    for k in range(3):
        P.ref_width[k], P.ref_height[k] = W, H
    return P


@pytest.mark.parametrize("W,H,bd,n_dec,rounds", [(2560, 1440, 8, 8, 40), (1280, 720, 10, 4, 60), (640, 368, 8, 12, 100)])
def test_decoders_in_one_process(hip, W, H, bd, n_dec, rounds):
    import refframe
    import workload
    rng = np.random.default_rng(77)
    dt = np.uint16 if bd > 8 else np.uint8
    dims, _ = refframe.plane_dims(W, H)
    refs = [[np.ascontiguousarray(workload.smooth_noise(rng, d[1], d[0], bd, sigma=1.5 + k).astype(dt)) for d in dims]
            for k in range(3)]
    frames = []
    for _ in range(2):
        blocks = blockgen.gen_blocks(rng, W, H, hip.BLOCK_DTYPE, intra_frac=0.08, skip_frac=0.35)
        coef, eob = blockgen.gen_coeffs(rng, blocks, W, H, bd)
        frames.append((blocks, coef, eob))
    P = _params(hip, W, H, bd)
    th = hip.LfThresh()
    hip.lib().vp9hip_lf_frame_init(32, 0, None, None, 0, 0, None, None, None, ctypes.byref(th))
    ALL = hip.PHASE_INTER | hip.PHASE_INTRA | hip.PHASE_LF
    want = []
    for blocks, coef, eob in frames:
        rf = refframe.RefFrame(refframe.load_ref(), blocks, W, H, bd, refs, [(W, H)] * 3, coef, eob, tiles=2)
        rf.run()
        want.append(refframe.frame_md5(rf.planes(), W, H))

    decs = []
    for s in range(n_dec):
        d = hip.Decoder(0)
        for k in range(3):
            d.upload(k, refs[k], W, H, bd)
        d.alloc_slot(3, W, H, bd)
        fr = frames[s % 2]
        d.begin_frame(P, fr[0], fr[2], fr[1])
        d.set_timing(False)
        decs.append(d)
    failures = []

    def drive(d):  # one host thread per decoder (ctypes drops the GIL): `rounds` frames queued back to back
        try:
            for _ in range(rounds):
                d.run(ALL, (0, 1, 2), 3, thresh=th)
            d.sync()  # raises if a filter row gave up waiting
        except Exception as e:  # noqa: BLE001
            failures.append(str(e))

    thr = [threading.Thread(target=drive, args=(d,)) for d in decs]
    for x in thr:
        x.start()
    for x in thr:
        x.join()
    assert not failures, failures[0]
    for s, d in enumerate(decs):
        got = [np.zeros((dd[1], dd[0]), dt) for dd in dims]
        d.download(3, got, W, H, bd)
        assert refframe.frame_md5(got, W, H) == want[s % 2], f"decoder {s}"
        d.close()


def test_first_launch_of_a_context_on_a_busy_gpu(hip):
    """A context's FIRST fused launch sets up its hand-off granules, ticket counters and error record.  Those fills
    must be ordered with the launch (they go into the context's own stream): the streams do not synchronise with the
    null stream, and a fill that landed after the launch had started — which takes a busy GPU, i.e. other decoders —
    wiped granules a row below was waiting for ("gave up waiting for the rows handed down by the row above", seen
    once in round 3's full run).  Two decoders keep the GPU busy; fresh decoders come and go beside them."""
    import refframe
    import workload
    W, H, bd = 1280, 720, 8
    rng = np.random.default_rng(5)
    dims, _ = refframe.plane_dims(W, H)
    refs = [[np.ascontiguousarray(workload.smooth_noise(rng, d[1], d[0], bd, sigma=1.5 + k).astype(np.uint8)) for d in dims]
            for k in range(3)]
    blocks = blockgen.gen_blocks(rng, W, H, hip.BLOCK_DTYPE, intra_frac=0.08, skip_frac=0.35)
    coef, eob = blockgen.gen_coeffs(rng, blocks, W, H, bd)
    P = _params(hip, W, H, bd)
    th = hip.LfThresh()
    hip.lib().vp9hip_lf_frame_init(32, 0, None, None, 0, 0, None, None, None, ctypes.byref(th))
    ALL = hip.PHASE_INTER | hip.PHASE_INTRA | hip.PHASE_LF
    rf = refframe.RefFrame(refframe.load_ref(), blocks, W, H, bd, refs, [(W, H)] * 3, coef, eob, tiles=2)
    rf.run()
    want = refframe.frame_md5(rf.planes(), W, H)

    def fresh():
        d = hip.Decoder(0)
        for k in range(3):
            d.upload(k, refs[k], W, H, bd)
        d.alloc_slot(3, W, H, bd)
        d.begin_frame(P, blocks, eob, coef)
        d.set_timing(False)
        return d

    stop = threading.Event()
    failures = []

    def busy(d):
        try:
            while not stop.is_set():
                for _ in range(20):
                    d.run(ALL, (0, 1, 2), 3, thresh=th)
                d.sync()
        except Exception as e:  # noqa: BLE001
            failures.append(str(e))

    background = [fresh() for _ in range(2)]
    thr = [threading.Thread(target=busy, args=(d,)) for d in background]
    for x in thr:
        x.start()
    try:
        for k in range(12):
            d = fresh()
            d.run(ALL, (0, 1, 2), 3, thresh=th)  # first launch of this context
            d.sync()
            got = [np.zeros((dd[1], dd[0]), np.uint8) for dd in dims]
            d.download(3, got, W, H, bd)
            assert refframe.frame_md5(got, W, H) == want, f"fresh decoder {k}"
            d.close()
    finally:
        stop.set()
        for x in thr:
            x.join()
    assert not failures, failures[0]
    for d in background:
        d.close()


@pytest.mark.parametrize("name,n_proc,threads", [("S-1440", 4, 4), ("s352_arf", 5, 1)])
def test_decoder_processes_side_by_side(name, n_proc, threads):
    exe = os.path.join(ROOT, "cuda-vp9_amd", "vp9hip_dec")
    for base in (os.path.join(ROOT, "tests", "streams_big"), os.path.join(ROOT, "tests", "golden", "streams")):
        ivf, gold = os.path.join(base, name + ".ivf"), os.path.join(base, name + ".md5")
        if os.path.exists(ivf):
            break
    else:
        pytest.skip(f"{name}.ivf absent (tests/golden/streams/make_streams.py --big)")
    if not os.path.exists(exe):
        pytest.skip("cuda-vp9_amd/vp9hip_dec not built")
    want = [l.strip() for l in open(gold) if l.strip()]
    loops = 3
    procs = [subprocess.Popen([exe, "--md5", "-o", "img-%wx%h-%4.i420", f"--loops={loops}", f"--threads={threads}", ivf],
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE) for _ in range(n_proc)]
    for i, p in enumerate(procs):
        out, err = p.communicate(timeout=600)
        err = err.decode(errors="replace")
        assert p.returncode == 0, f"process {i}: rc {p.returncode}: {err[-400:]}"
        assert "gave up" not in err, f"process {i}: {err[-400:]}"
        lines = [l.strip() for l in out.decode().splitlines() if l.strip()]
        assert len(lines) == loops * len(want) and sorted(set(lines)) == sorted(set(want)), \
            f"process {i}: MD5 lines differ from {gold}"
