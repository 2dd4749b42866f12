"""GPU parity THROUGH THE REFERENCE'S CALL SURFACE: wrap_cuda_inter_prediction /
wrap_cuda_intra_prediction (shim/vp9hip_libvpx_shim.c) are called by a harness that hands them the
reference's own structures (VP9Decoder / VP9_COMMON / BufferPool frame buffers from
vpx_realloc_frame_buffer / MODE_INFO + ModeInfoBuf / frameBuf laid out by initBuf's rules), the way
decode_tiles does (libvpx/vp9/decoder/vp9_decodeframe.c:2546, 2564).  The delivered host frame
must equal the oracle's sequential reconstruction of the same blocks, bit for bit, in both
residual modes (GPU inverse transforms from dqcoeff/plane_eob; CPU-transformed int64 planes)."""
import ctypes
import os

import numpy as np
import pytest

import blockgen
from frame_check import OFrame

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def harness():
    so = os.path.join(ROOT, "shim", "build", "libshimtest.so")
    if not os.path.exists(so):
        pytest.fail(f"{so} missing: run `make -C shim` where /root/reference exists (the built .so travels)")
    lib = ctypes.CDLL(so)
    lib.shimtest_create.restype = ctypes.c_void_p
    lib.shimtest_destroy.argtypes = [ctypes.c_void_p]
    return lib


def _dims(W, H):
    aw, ah = (W + 7) & ~7, (H + 7) & ~7
    return ([(aw, ah), (aw // 2, ah // 2), (aw // 2, ah // 2)],
            [(W, H), ((W + 1) // 2, (H + 1) // 2), ((W + 1) // 2, (H + 1) // 2)])


def _oframe(planes, dims, crop, bd):
    f = OFrame()
    for p, a in enumerate(planes):
        f.plane[p], f.stride[p] = a.ctypes.data, a.shape[1]
        f.width[p], f.height[p] = crop[p]
        f.awidth[p], f.aheight[p] = dims[p]
    f.bit_depth, f.hbd = bd, int(bd > 8)
    return f


def _expected(hip, oracle, blocks, W, H, bd, ref_sizes, refs, coef, eob, tiles, lossless):
    """Sequential oracle reconstruction of the frame from the product's packed lists (the packer
    itself is pinned against the reference's objects in test_packer_vs_ref.py)."""
    P = hip.FrameParams()
    P.width, P.height, P.ss_x, P.ss_y, P.bit_depth, P.hbd = W, H, 1, 1, bd, int(bd > 8)
    P.lossless, P.log2_tile_cols = lossless, tiles
    for k, (rw, rh) in enumerate(ref_sizes):
        P.ref_width[k], P.ref_height[k] = rw, rh
    pk = hip.Packer()
    L = pk.pack(P, blocks, eob)
    pk.close()
    assert L["coeff_count"] == [len(c) for c in coef]
    allc = np.concatenate(coef + [np.zeros(16, np.int32)])
    cp = allc.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))
    dims, crop = _dims(W, H)
    dt = np.uint16 if bd > 8 else np.uint8
    out = [np.zeros((d[1], d[0]), dt) for d in dims]
    dst = _oframe(out, dims, crop, bd)
    rarr = (OFrame * 3)()
    for k, (rw, rh) in enumerate(ref_sizes):
        rd, rc = _dims(rw, rh)
        rarr[k] = _oframe(refs[k], rd, rc, bd)
    t = L["inter_tasks"]
    if len(t):
        oracle.vp9o_recon_inter_list(t.ctypes.data_as(ctypes.c_void_p), len(t), rarr, ctypes.byref(dst))
    t = L["txb"]
    if len(t):
        oracle.vp9o_recon_txb_list(t.ctypes.data_as(ctypes.c_void_p), len(t), cp, ctypes.byref(dst))
    t = L["intra_decode_order"]
    if len(t):
        oracle.vp9o_recon_intra_list(t.ctypes.data_as(ctypes.c_void_p), len(t), cp, ctypes.byref(dst))
    return out, L, allc


def _residual_planes(oracle, L, allc, dims, bd, lossless):
    """What the reference's CPU phase B leaves in frameBuf.plane_residuals: the inverse transform of
    every coded block (post-shift, pre-clip), zero elsewhere."""
    res = [np.zeros((d[1], d[0]), np.int64) for d in dims]
    tmp = np.zeros(1024, np.int32)
    for recs, is_intra in ((L["txb"], False), (L["intra_decode_order"], True)):
        for r in recs:
            if r["eob"] == 0:
                continue
            n = 4 << int(r["tx_size"])
            c = np.ascontiguousarray(allc[int(r["coeff_off"]):int(r["coeff_off"]) + n * n])
            oracle.vp9o_inv_txfm_residual(n, int(r["tx_type"]) & 3, int(lossless), int(bd > 8),
                                          c.ctypes.data_as(ctypes.c_void_p), tmp.ctypes.data_as(ctypes.c_void_p))
            pl, x, y = int(r["plane"]), int(r["x"]), int(r["y"])
            vw, vh = min(n, dims[pl][0] - x), min(n, dims[pl][1] - y)
            res[pl][y:y + vh, x:x + vw] = tmp[:n * n].reshape(n, n)[:vh, :vw]
    return res


def _call(harness, h, recs, W, H, bd, tiles, lossless, inter, coefficient_mode, refs, ref_sizes, coef, eob, res, got, opts=None):
    flat_refs = [a for r in refs for a in r]
    ref_ptrs = (ctypes.c_void_p * 9)(*[a.ctypes.data for a in flat_refs])
    rw = (ctypes.c_int * 3)(*[s[0] for s in ref_sizes])
    rh = (ctypes.c_int * 3)(*[s[1] for s in ref_sizes])
    dq = (ctypes.c_void_p * 3)(*[c.ctypes.data if len(c) else None for c in coef])
    eob_keep = [np.ascontiguousarray(e) for e in eob]
    eobp = (ctypes.c_void_p * 3)(*[e.ctypes.data for e in eob_keep])
    resp = (ctypes.c_void_p * 3)(*[r.ctypes.data for r in res]) if res is not None else None
    outp = (ctypes.c_void_p * 3)(*[g.ctypes.data for g in got])
    times = (ctypes.c_double * 4)()
    err = ctypes.create_string_buffer(512)
    o = (ctypes.c_int32 * 8)(*opts) if opts is not None else None
    rc = harness.shimtest_frame(ctypes.c_void_p(h), recs.ctypes.data_as(ctypes.c_void_p), len(recs), W, H, bd, int(bd > 8),
                                tiles, int(lossless), int(inter), int(coefficient_mode), ref_ptrs, rw, rh, dq, eobp, resp,
                                outp, times, err, 512, o)
    return rc, err.value.decode(), list(times)


def _run(harness, hip, oracle, W, H, bd, seed, *, inter=True, coefficient_mode=True, tiles=0, lossless=False,
         ref_sizes=None, gen_kw=None):
    rng = np.random.default_rng(seed)
    dt = np.uint16 if bd > 8 else np.uint8
    kw = dict(gen_kw or {})
    if not inter:
        kw["all_intra"] = True
    blocks = blockgen.gen_blocks(rng, W, H, hip.BLOCK_DTYPE, **kw)
    if lossless:
        blocks["tx_size"] = 0
    ref_sizes = ref_sizes or [(W, H)] * 3
    dims, crop = _dims(W, H)
    refs = []
    for (rw, rh) in ref_sizes:
        rd, _ = _dims(rw, rh)
        refs.append([np.ascontiguousarray(blockgen_noise(rng, d[1], d[0], bd)).astype(dt) for d in rd])
    coef, eob = blockgen.gen_coeffs(rng, blocks, W, H, bd, lossless=lossless)
    expect, L, allc = _expected(hip, oracle, blocks, W, H, bd, ref_sizes, refs, coef, eob, tiles, int(lossless))
    res = _residual_planes(oracle, L, allc, dims, bd, lossless) if not coefficient_mode else None

    recs = blockgen.to_ref_records(blocks)
    got = [np.zeros((d[1], d[0]), dt) for d in dims]
    h = harness.shimtest_create()
    rc, err, times = _call(harness, h, recs, W, H, bd, tiles, lossless, inter, coefficient_mode, refs, ref_sizes, coef, eob,
                           res, got)
    harness.shimtest_destroy(ctypes.c_void_p(h))
    return rc, err, got, expect, times, L


def blockgen_noise(rng, h, w, bd):
    import workload
    return workload.smooth_noise(rng, h, w, bd, sigma=1.5)


def _assert_equal(got, expect):
    for p in range(3):
        bad = np.argwhere(got[p] != expect[p])
        assert bad.size == 0, f"plane {p}: {len(bad)} samples differ, first at {bad[:5].tolist()}"


@pytest.mark.parametrize("W,H,bd,kw", [
    (352, 288, 8, {}),
    (330, 250, 10, dict(tiles=1)),
    (200, 136, 8, dict(gen_kw=dict(intra_frac=0.5))),
    (256, 192, 8, dict(ref_sizes=[(300, 200), (256, 192), (128, 96)])),
    (256, 256, 8, dict(inter=False)),                       # key frame: only the intra wrapper is called
    (192, 128, 12, dict(inter=False, tiles=1)),
    (128, 128, 8, dict(lossless=True)),
    (1920, 1080, 8, dict(tiles=2)),
])
def test_wrappers_match_oracle_coefficient_mode(harness, hip, oracle, W, H, bd, kw):
    rc, err, got, expect, times, L = _run(harness, hip, oracle, W, H, bd, seed=W + H + bd, **kw)
    assert rc == 0, err
    _assert_equal(got, expect)
    assert times[3] > 0 and all(t >= 0 for t in times)


@pytest.mark.parametrize("W,H,bd,kw", [
    (352, 288, 10, {}),
    (200, 136, 12, dict(gen_kw=dict(intra_frac=0.5), tiles=1)),
    (256, 256, 10, dict(inter=False)),
    (640, 360, 10, dict(gen_kw=dict(compound_frac=0.5))),
])
def test_wrappers_match_oracle_residual_plane_mode(harness, hip, oracle, W, H, bd, kw):
    """The unchanged reference's contract: its CPU phase B has left int64 residual planes."""
    rc, err, got, expect, times, L = _run(harness, hip, oracle, W, H, bd, seed=W + H + bd, coefficient_mode=False, **kw)
    assert rc == 0, err
    _assert_equal(got, expect)


def test_residual_plane_mode_refuses_8bit_frames(harness, hip, oracle):
    rc, err, *_ = _run(harness, hip, oracle, 128, 128, 8, seed=5, coefficient_mode=False)
    assert rc != 0
    assert "high-bitdepth" in err


def test_error_unwinds_through_libvpx_trap_and_the_decoder_stays_usable(harness, hip, oracle):
    """A failing wrapper reports through vpx_internal_error, which longjmps to the trap libvpx installs around a
    frame (the harness installs the same one): the call stack of the wrapper is abandoned half-way.  Everything
    the shim owns lives in its per-decoder state, so the SAME decoder instance must decode the next, valid
    frame correctly — nothing leaked, no half-open frame left behind."""
    W, H, bd = 192, 128, 8
    rng = np.random.default_rng(77)
    dims, crop = _dims(W, H)
    blocks = blockgen.gen_blocks(rng, W, H, hip.BLOCK_DTYPE)
    refs = [[np.ascontiguousarray(blockgen_noise(rng, d[1], d[0], bd)).astype(np.uint8) for d in dims] for _ in range(3)]
    coef, eob = blockgen.gen_coeffs(rng, blocks, W, H, bd)
    expect, L, allc = _expected(hip, oracle, blocks, W, H, bd, [(W, H)] * 3, refs, coef, eob, 0, 0)
    recs = blockgen.to_ref_records(blocks)
    h = harness.shimtest_create()
    for attempt in range(3):
        # residual-plane mode on an 8-bit frame: refused inside the inter wrapper, after the state was touched
        res = [np.zeros((d[1], d[0]), np.int64) for d in dims]
        got = [np.zeros((d[1], d[0]), np.uint8) for d in dims]
        rc, err, _ = _call(harness, h, recs, W, H, bd, 0, False, True, False, refs, [(W, H)] * 3, coef, eob, res, got)
        assert rc != 0 and "high-bitdepth" in err
        got = [np.zeros((d[1], d[0]), np.uint8) for d in dims]
        rc, err, _ = _call(harness, h, recs, W, H, bd, 0, False, True, True, refs, [(W, H)] * 3, coef, eob, None, got)
        assert rc == 0, err
        _assert_equal(got, expect)
    harness.shimtest_destroy(ctypes.c_void_p(h))


def _lf_oracle(hip, oracle, planes, L, dims, bd, sharp):
    """Loop filter of `planes` in place through the oracle, masks from the C packer (pinned equal to
    vp9_build_mask + vp9_adjust_mask), thresholds from vp9hip_lf_frame_init (pinned equal to libvpx's)."""
    from frame_check import OThresh
    th = hip.LfThresh()
    hip.lib().vp9hip_lf_frame_init(32, sharp, None, None, 0, 0, None, None, None, ctypes.byref(th))
    oth = OThresh.from_buffer_copy(bytes(th))
    PAD = 16
    bufs = [np.zeros((d[1] + PAD, d[0] + PAD), planes[0].dtype) for d in dims]
    for b, p in zip(bufs, planes):
        b[:p.shape[0], :p.shape[1]] = p
    ptrs = (ctypes.c_void_p * 3)(*[b.ctypes.data for b in bufs])
    strides = (ctypes.c_int * 3)(*[b.shape[1] for b in bufs])
    oracle.vp9o_loop_filter_frame(L["lfm"].ctypes.data_as(ctypes.c_void_p), L["sb_rows"], L["sb_cols"], ctypes.byref(oth),
                                  ptrs, strides, dims[0][1] // 8, bd, int(bd > 8), 3)
    return [b[:d[1], :d[0]].copy() for b, d in zip(bufs, dims)]


@pytest.mark.parametrize("W,H,bd,sharp", [(352, 288, 8, 0), (330, 250, 10, 4), (640, 360, 8, 0)])
def test_gpu_loop_filter_and_resident_references_through_the_wrappers(harness, hip, oracle, W, H, bd, sharp):
    """vp9hip_shim_set_gpu_loop_filter: the intra wrapper also runs phase E — masks built by the C packer
    (pinned equal to the reference's vp9_build_mask + vp9_adjust_mask in tests/test_packer_vs_ref.py; inter
    blocks whose eobs are all zero count as skipped, as in stock libvpx) — and the decoded frame stays in the
    device pool: the next frame references it while the HOST copy of that buffer is overwritten with garbage."""
    rng = np.random.default_rng(W + bd)
    dt = np.uint16 if bd > 8 else np.uint8
    dims, crop = _dims(W, H)
    sizes = [(W, H)] * 3
    refs = [[np.ascontiguousarray(blockgen_noise(rng, d[1], d[0], bd)).astype(dt) for d in dims] for _ in range(3)]
    h = harness.shimtest_create()
    # ---- frame A into frame buffer 5, references uploaded from buffers 0, 1, 2
    blocksA = blockgen.gen_blocks(rng, W, H, hip.BLOCK_DTYPE, levels=(0, 10, 32, 50))
    coefA, eobA = blockgen.gen_coeffs(rng, blocksA, W, H, bd)
    preA, LA, _ = _expected(hip, oracle, blocksA, W, H, bd, sizes, refs, coefA, eobA, 0, 0)
    P = hip.FrameParams()
    P.width, P.height, P.ss_x, P.ss_y, P.bit_depth, P.hbd, P.build_lf_masks = W, H, 1, 1, bd, int(bd > 8), 1
    for k in range(3):
        P.ref_width[k], P.ref_height[k] = W, H
    pk = hip.Packer()
    LmA = pk.pack(P, blocksA, eobA)
    expA = _lf_oracle(hip, oracle, preA, LmA, dims, bd, sharp)
    gotA = [np.zeros((d[1], d[0]), dt) for d in dims]
    rc, err, _ = _call(harness, h, blockgen.to_ref_records(blocksA), W, H, bd, 0, False, True, True, refs, sizes, coefA, eobA,
                       None, gotA, opts=(1, sharp, 5, 0, 1, 2, 7, 0))
    assert rc == 0, err
    _assert_equal(gotA, expA)
    assert any((a != b).any() for a, b in zip(expA, preA))  # the filter did change the frame
    # ---- frame B into buffer 6: LAST = buffer 5 (frame A, NOT re-filled, host copy poisoned), GOLDEN = 1
    blocksB = blockgen.gen_blocks(rng, W, H, hip.BLOCK_DTYPE, intra_frac=0.1, levels=(0, 20, 40))
    coefB, eobB = blockgen.gen_coeffs(rng, blocksB, W, H, bd)
    refsB = [expA, refs[1], refs[2]]
    preB, LB, _ = _expected(hip, oracle, blocksB, W, H, bd, sizes, refsB, coefB, eobB, 0, 0)
    LmB = pk.pack(P, blocksB, eobB)
    expB = _lf_oracle(hip, oracle, preB, LmB, dims, bd, sharp)
    gotB = [np.zeros((d[1], d[0]), dt) for d in dims]
    rc, err, _ = _call(harness, h, blockgen.to_ref_records(blocksB), W, H, bd, 0, False, True, True, refsB, sizes, coefB, eobB,
                       None, gotB, opts=(1, sharp, 6, 5, 1, 2, 0b110, 0b001))
    assert rc == 0, err
    _assert_equal(gotB, expB)
    pk.close()
    harness.shimtest_destroy(ctypes.c_void_p(h))
