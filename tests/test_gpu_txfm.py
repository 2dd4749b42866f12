"""GPU parity: vp9hip_idct_add_batch vs the oracle (bit-exact), all sizes / types / eob classes."""
import numpy as np
import pytest

from vp9ref import i32p, u8p, u16p

pytestmark = pytest.mark.gpu


def make_coeffs(rng, n, eob_class, lim, kind):
    c = np.zeros((n, n), np.int32)
    k = {0: 1, 1: min(n, 4), 2: n}[eob_class]
    if kind == 0:
        c[:k, :k] = rng.integers(-lim, lim, (k, k))
    elif kind == 1:
        c[:k, :k] = rng.integers(-lim // 64, lim // 64 + 1, (k, k))
    else:
        c[rng.integers(0, k), rng.integers(0, k)] = rng.choice([-lim, lim - 1])
    eob = {0: 1, 1: min(n * n, 10), 2: n * n}[eob_class]
    return c, eob


@pytest.mark.parametrize("bd,hbd", [(8, False), (10, True), (12, True), (8, True)])
def test_idct_add_batch_matches_oracle(hip, oracle, bd, hbd):
    rng = np.random.default_rng(100 + bd + hbd)
    W, H = 200, 136  # not a multiple of 64: blocks at the right/bottom edge get clipped
    ctx = hip.Context(0)
    frame = hip.DevFrame(ctx, W, H, bit_depth=bd, hbd=hbd)
    dt = np.uint16 if hbd else np.uint8
    planes = [rng.integers(0, 1 << bd, (d[3], d[2])).astype(dt) for d in frame.dims]
    frame.upload(planes)
    expect = [p.copy() for p in planes]
    recs, coeffs, off = [], [], 0
    lim = 1 << (bd + 8) if hbd else 32768
    for plane in range(3):
        aw, ah = frame.dims[plane][2], frame.dims[plane][3]
        occupied = np.zeros((ah // 4 + 16, aw // 4 + 16), bool)
        for it in range(500):
            ts = int(rng.integers(0, 4))
            n = 4 << ts
            x = int(rng.integers(0, aw // 4)) * 4
            y = int(rng.integers(0, ah // 4)) * 4
            if occupied[y // 4:y // 4 + n // 4, x // 4:x // 4 + n // 4].any():
                continue
            occupied[y // 4:y // 4 + n // 4, x // 4:x // 4 + n // 4] = True
            lossless = ts == 0 and it % 11 == 0
            tx_type = int(rng.integers(0, 4)) if (ts < 3 and plane == 0 and not lossless) else 0
            c, eob = make_coeffs(rng, n, it % 3, lim, (it // 3) % 3)
            recs.append((off, x, y, plane, ts, tx_type | (0x80 if lossless else 0), 0, eob, 0))
            coeffs.append(c.ravel())
            # oracle on a padded copy so that out-of-frame rows/cols are simply dropped
            pad = np.zeros((ah + 64, aw + 64), dt)
            pad[:ah, :aw] = expect[plane]
            sub = pad[y:, x:]
            if hbd:
                blk = np.ascontiguousarray(sub[:n, :n])
                oracle.vp9o_highbd_inv_txfm_add(n, tx_type, int(lossless), i32p(c), u16p(blk), n, eob, bd)
            else:
                blk = np.ascontiguousarray(sub[:n, :n])
                oracle.vp9o_inv_txfm_add(n, tx_type, int(lossless), i32p(c), u8p(blk), n, eob)
            pad[y:y + n, x:x + n] = blk
            expect[plane] = pad[:ah, :aw].copy()
            off += n * n
    blocks = np.array(recs, dtype=hip.TXB_DTYPE)
    blocks, counts = hip.sort_txb_by_size(blocks)
    d_blocks = ctx.alloc(blocks)
    d_coeffs = ctx.alloc(np.concatenate(coeffs).astype(np.int32))
    ctx.idct_add_batch(d_blocks, counts, d_coeffs, frame)
    ctx.sync()
    got = frame.download()
    for p in range(3):
        assert np.array_equal(got[p], expect[p]), f"plane {p} differs at {np.argwhere(got[p] != expect[p])[:5]}"
    assert sum(counts) > 300
    ctx.close()
