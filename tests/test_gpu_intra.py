"""GPU parity: vp9hip_intra_pred_waves (wave-ordered predict + residual) vs the oracle run
sequentially in decode order."""
import ctypes

import numpy as np
import pytest

import synth
from vp9ref import i32p, u8p, u16p, ptr_at

pytestmark = pytest.mark.gpu


class IntraArgs(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int) for n in
                ("mode", "bs", "have_top", "have_left", "have_right", "x", "y", "frame_width", "frame_height")]


def build_case(rng, W, H, bd, hbd, with_residual):
    aw, ah = (W + 7) & ~7, (H + 7) & ~7
    dims = [(aw, ah), (aw // 2, ah // 2), (aw // 2, ah // 2)]
    tasks, coeffs, off = [], [], 0
    lim = (1 << (bd + 8)) if hbd else 32768
    for (bx, by, bsz) in synth.quad_blocks(rng, aw, ah):
        tx_log2 = int(rng.integers(2, min(5, int(np.log2(bsz))) + 1))
        if bsz == 8 and rng.random() < 0.5:
            tx_log2 = 2
        blk_mode = int(rng.integers(0, 10))
        uv_mode = int(rng.integers(0, 10))
        for plane in range(3):
            ss = 1 if plane else 0
            pw = bsz >> ss
            tlog = tx_log2 if plane == 0 else min(tx_log2, int(np.log2(pw)))
            bs = 1 << tlog
            px, py = bx >> ss, by >> ss
            paw, pah = dims[plane]
            for ty in range(0, pw, bs):
                for tx in range(0, pw, bs):
                    x, y = px + tx, py + ty
                    if x >= paw or y >= pah:
                        continue
                    mode = (int(rng.integers(0, 10)) if (plane == 0 and bsz == 8 and bs == 4) else
                            (blk_mode if plane == 0 else uv_mode))
                    t = dict(plane=plane, x=x, y=y, bs=bs, tx_size=tlog - 2, mode=mode,
                             have_top=int(y > 0), have_left=int(x > 0), have_right=int(tx + bs < pw),
                             eob=0, coeff_off=0, tx_type=0)
                    if with_residual and rng.random() < 0.7:
                        n = bs
                        c = np.zeros((n, n), np.int32)
                        kind = int(rng.integers(0, 3))
                        if kind == 0:
                            c[0, 0] = int(rng.integers(-lim // 8, lim // 8))
                            t["eob"] = 1
                        else:
                            k = n if kind == 1 else min(n, 4)
                            c[:k, :k] = rng.integers(-lim // 32, lim // 32 + 1, (k, k))
                            t["eob"] = n * n if kind == 1 else 10
                        if plane == 0 and n < 32:
                            t["tx_type"] = int(rng.integers(0, 4))
                        t["coeff_off"] = off
                        coeffs.append(c.ravel())
                        off += n * n
                        t["c"] = c
                    tasks.append(t)
    return dims, tasks, (np.concatenate(coeffs).astype(np.int32) if coeffs else np.zeros(16, np.int32))


@pytest.mark.parametrize("bd,hbd,resid", [(8, False, False), (8, False, True), (10, True, True), (12, True, False)])
def test_intra_waves_match_sequential_oracle(hip, oracle, bd, hbd, resid):
    rng = np.random.default_rng(300 + bd + resid)
    W, H = 200, 136
    dims, tasks, coeffs = build_case(rng, W, H, bd, hbd, resid)
    dt = np.uint16 if hbd else np.uint8
    ctx = hip.Context(0)
    frame = hip.DevFrame(ctx, W, H, bit_depth=bd, hbd=hbd)
    planes = [rng.integers(0, 1 << bd, (ah, aw)).astype(dt) for (aw, ah) in dims]
    frame.upload(planes)
    # sequential oracle in decode order, on padded planes (overhanging blocks write into padding)
    pads = []
    for p, (aw, ah) in enumerate(dims):
        pad = np.zeros((ah + 64, aw + 64), dt)
        pad[:ah, :aw] = planes[p]
        pads.append(pad)
    for t in tasks:
        pad = pads[t["plane"]]
        aw, ah = dims[t["plane"]]
        a = IntraArgs(t["mode"], t["bs"], t["have_top"], t["have_left"], t["have_right"], t["x"], t["y"], aw, ah)
        pp = ptr_at(pad, t["y"], t["x"])
        if hbd:
            oracle.vp9o_highbd_predict_intra(ctypes.byref(a), pp, pad.shape[1], pp, pad.shape[1], bd)
            if t["eob"]:
                oracle.vp9o_highbd_inv_txfm_add(t["bs"], t["tx_type"], 0, i32p(t["c"]), pp, pad.shape[1], t["eob"], bd)
        else:
            oracle.vp9o_predict_intra(ctypes.byref(a), pp, pad.shape[1], pp, pad.shape[1])
            if t["eob"]:
                oracle.vp9o_inv_txfm_add(t["bs"], t["tx_type"], 0, i32p(t["c"]), pp, pad.shape[1], t["eob"])
    expect = [pads[p][:ah, :aw] for p, (aw, ah) in enumerate(dims)]
    # GPU: sort by dependency level
    levels = synth.intra_levels(tasks, dims)
    order = np.argsort(levels, kind="stable")
    recs = np.zeros(len(tasks), hip.INTRA_DTYPE)
    for i, j in enumerate(order):
        t = tasks[j]
        recs[i] = (t["coeff_off"], t["x"], t["y"], t["plane"], t["tx_size"], t["tx_type"], t["mode"], t["eob"],
                   t["have_top"] | (t["have_left"] << 1) | (t["have_right"] << 2), 0)
    sl = levels[order]
    nw = int(sl.max())
    wave_start = np.searchsorted(sl, np.arange(1, nw + 2)).astype(np.int32)
    d_tasks = ctx.alloc(recs)
    d_coeffs = ctx.alloc(coeffs)
    ctx.intra_pred_waves(d_tasks, wave_start, d_coeffs, frame)
    ctx.sync()
    got = frame.download()
    for p in range(3):
        bad = np.argwhere(got[p] != expect[p])
        assert bad.size == 0, f"plane {p}: {len(bad)} px differ, first {bad[:4]} (waves={nw})"
    assert len(tasks) > 200 and nw > 5
    ctx.close()


def test_intra_islands_match_sequential_oracle(hip, oracle):
    """Island form (one launch, one workgroup per connected component) on an inter frame with
    intra clusters and on an all-intra frame: same pixels as the oracle run in decode order."""
    import cuda_vp9_amd.pipeline as pipeline
    import workload
    import frame_check
    for kw in (dict(intra_frac=0.4), dict(all_intra=True)):
        wl = workload.make_frame_workload(328, 200, seed=11, **kw)
        assert len(wl["intra_islands"]) >= 2
        assert kw.get("all_intra") or len(wl["intra_big_tasks"]) == 0  # clusters: islands only
        ctx = hip.Context(0)
        job = pipeline.FrameJob(ctx, wl)
        for use_islands in (True, False):
            job.use_islands = use_islands
            job.clear_dst()
            job.run(phases=("inter", "txb", "intra"))
            ctx.sync()
            got = job.download()
            exp, _ = frame_check.oracle_frame(oracle, wl, phases=("inter", "txb", "intra"))
            for p in range(3):
                assert np.array_equal(got[p], exp[p]), (kw, use_islands, p)
        job.free()
        ctx.close()
