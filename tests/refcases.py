"""Seeded case generators shared by the golden-vector generator (tests/golden/make_golden.py),
the oracle-vs-reference tests and the GPU rtcd-twin tests.  Input distributions follow the
reference's own tests: energy-bounded / extreme single coefficients as in
test/partial_idct_test.cc:126-139, 252-291; random + high-contrast planes as in
test/convolve_test.cc:900-1080; smooth + noisy edges as in test/lpf_test.cc:174-420."""
import numpy as np

TXFM_VARIANTS = {4: [1, 16], 8: [1, 12, 64], 16: [1, 10, 38, 256], 32: [1, 34, 135, 1024]}
CONV_NAMES = {0: "convolve_copy", 4: "convolve_avg", 1: "convolve8_horiz", 2: "convolve8_vert", 3: "convolve8",
              5: "convolve8_avg_horiz", 6: "convolve8_avg_vert", 7: "convolve8_avg"}
INTRA_NAMES = {"dc": 0, "v": 1, "h": 2, "d45": 3, "d135": 4, "d117": 5, "d153": 6, "d207": 7, "d63": 8, "tm": 9,
               "dc_128": 10, "dc_left": 11, "dc_top": 12}


def txfm_coeffs(rng, n, eob_variant, kind, lim):
    c = np.zeros((n, n), np.int32)
    k = 1 if eob_variant == 1 else {12: 4, 10: 4, 38: 8, 34: 8, 135: 16}.get(eob_variant, n)
    if kind == 0:
        c[:k, :k] = rng.integers(-lim, lim, (k, k))
    elif kind == 1:
        c[:k, :k] = rng.integers(-lim // 64, lim // 64 + 1, (k, k))
    else:
        c[rng.integers(0, k), rng.integers(0, k)] = rng.choice([-lim, lim - 1])
    return c


def conv_case(rng, it):
    w = int(rng.choice([4, 8, 16, 32, 64]))
    h = int(rng.choice([4, 8, 16, 32, 64]))
    filt = int(rng.integers(0, 5))
    mode = int(rng.choice(list(CONV_NAMES)))
    scaled = it % 5 == 0
    xs = int(rng.choice([16, 20, 24, 32])) if scaled else 16
    ys = int(rng.choice([16, 20, 24, 32])) if scaled else 16
    x0, y0 = int(rng.integers(0, 16)), int(rng.integers(0, 16))
    hbd = it % 2
    bd = [8, 10, 12][it % 3] if hbd else 8
    H = W = 64 * 2 + 24
    dt = np.uint16 if hbd else np.uint8
    kind = it % 4
    if kind == 0:
        src = rng.integers(0, 1 << bd, (H, W)).astype(dt)
    elif kind == 1:
        src = (rng.integers(0, 2, (H, W)) * ((1 << bd) - 1)).astype(dt)
    else:
        src = np.clip(rng.normal((1 << bd) / 2, (1 << bd) / 6, (H, W)), 0, (1 << bd) - 1).astype(dt)
    dst = rng.integers(0, 1 << bd, (64, 80)).astype(dt)
    return dict(w=w, h=h, filt=filt, mode=mode, scaled=int(scaled), xs=xs, ys=ys, x0=x0, y0=y0, hbd=hbd, bd=bd,
                src=src, dst=dst)


def lpf_case(rng, it):
    vertical = it % 2
    kind = [4, 8, 16][(it // 2) % 3]
    dual = (it // 6) % 2
    hbd = (it // 12) % 2
    bd = [8, 10, 12][it % 3] if hbd else 8
    dt = np.uint16 if hbd else np.uint8
    base = rng.integers(0, 1 << bd)
    amp = int(rng.choice([1, 2, 4, 16, 64])) << (bd - 8)
    img = np.clip(base + rng.integers(-amp, amp + 1, (40, 40)), 0, (1 << bd) - 1).astype(dt)
    if it % 7 == 0:
        img = rng.integers(0, 1 << bd, (40, 40)).astype(dt)
    th = np.array([rng.integers(0, 256), rng.integers(0, 64), rng.integers(0, 16), rng.integers(0, 256),
                   rng.integers(0, 64), rng.integers(0, 16)], np.uint8)
    return dict(vertical=vertical, kind=kind, dual=dual, hbd=hbd, bd=bd, img=img, th=th)


def lpf_name(c):
    return "vpx_%slpf_%s_%d%s_c" % ("highbd_" if c["hbd"] else "", "vertical" if c["vertical"] else "horizontal",
                                    c["kind"], "_dual" if c["dual"] else "")
