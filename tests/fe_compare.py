"""Compares the product's bitstream front-end (vp9hip_fe, CPU only) with the reference's own parse of the same
stream: oracle/_ref/vpx/vpxdec_c with VP9_ORACLE_DUMP_BLOCKS writes one 64-byte record per block (the layout of
vp9hip_block); every field the reconstruction path reads is compared block by block, plus a checksum over each
block's eobs and coefficients.   python tests/fe_compare.py stream.ivf   prints the first difference."""
import ctypes
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

REC = np.dtype([("mi_row", "<i2"), ("mi_col", "<i2"), ("sb_type", "u1"), ("tx_size", "u1"), ("skip", "u1"),
                ("interp_filter", "u1"), ("ref_frame", "i1", (2,)), ("mode", "u1"), ("uv_mode", "u1"),
                ("sub_mode", "u1", (4,)), ("filter_level", "u1"), ("segment_id", "u1"), ("seg_pred", "u1"),
                ("skip_parsed", "u1"), ("mv", "<i2", (2, 2)), ("sub_mv", "<i2", (4, 2, 2)), ("checksum", "<u4")])
assert REC.itemsize == 64


def ivf_frames(path):
    import __graft_entry__ as g
    return g.load_pkg().ivf_packets(path)


def parse_stream(hip, path, threads=1):
    """Every decoded frame's block records as the product's front-end parses them."""
    os.environ["VP9HIP_FE_CHECKSUMS"] = "1"  # read at creation: per-block checksums over eobs + coefficients
    fe = hip.FrontEnd(threads=threads)
    del os.environ["VP9HIP_FE_CHECKSUMS"]
    frames = []
    try:
        for pkt in hip.ivf_packets(path):
            for data in fe.frames_of(pkt):
                fr = fe.parse(data)
                if fr.show_existing:
                    continue
                buf = (ctypes.c_char * (64 * fr.n_blocks)).from_address(fr.blocks)
                frames.append(np.frombuffer(bytes(buf), REC).copy())
    finally:
        fe.close()
    return frames


def reference_blocks(path):
    dec = os.path.join(ROOT, "oracle", "_ref", "vpx", "vpxdec_c")
    with tempfile.NamedTemporaryFile(suffix=".blocks", delete=False) as tmp:
        out = tmp.name
    try:
        subprocess.run([dec, "--noblit", path], env=dict(os.environ, VP9_ORACLE_DUMP_BLOCKS=out), stdout=subprocess.DEVNULL,
                       stderr=subprocess.DEVNULL, check=True, timeout=1200)
        data = open(out, "rb").read()
    finally:
        os.unlink(out)
    frames, pos = [], 0
    while pos + 16 <= len(data):
        magic, n, w, h = np.frombuffer(data[pos:pos + 16], "<i4")
        assert magic == 0x56503946
        pos += 16
        frames.append(np.frombuffer(data[pos:pos + 64 * n], REC).copy())
        pos += 64 * n
    return frames


def compare(mine, ref):
    """None when equal, else a description of the first difference."""
    if len(mine) != len(ref):
        return f"{len(mine)} decoded frames, the reference has {len(ref)}"
    for f, (a, b) in enumerate(zip(mine, ref)):
        if len(a) != len(b):
            n = min(len(a), len(b))
            bad = np.nonzero((a["mi_row"][:n] != b["mi_row"][:n]) | (a["mi_col"][:n] != b["mi_col"][:n]) | (a["sb_type"][:n] != b["sb_type"][:n]))[0]
            return f"frame {f}: {len(a)} blocks, the reference has {len(b)}; first differing position at block {bad[0] if len(bad) else n}"
        inter = b["ref_frame"][:, 0] > 0
        sub8 = b["sb_type"] < 3
        checks = [("mi_row", None), ("mi_col", None), ("sb_type", None), ("tx_size", None), ("skip_parsed", None), ("ref_frame", None),
                  ("segment_id", None), ("filter_level", None), ("mode", None), ("uv_mode", ~inter), ("interp_filter", inter),
                  ("sub_mode", ~inter & sub8), ("mv", inter), ("sub_mv", inter & sub8), ("checksum", None)]
        for name, mask in checks:
            x, y = a[name], b[name]
            if name == "ref_frame":  # the second entry: "none" is -1 there, <= 0 here
                x = np.where(x <= 0, np.int8(0) if False else x, x)
                ne = (x[:, 0] != y[:, 0]) | ((x[:, 1] > 0) != (y[:, 1] > 0)) | ((x[:, 1] > 0) & (x[:, 1] != y[:, 1]))
            elif name == "mv":
                second = (b["ref_frame"][:, 1] > 0)
                ne = (x[:, 0] != y[:, 0]).any(axis=1) | (second & (x[:, 1] != y[:, 1]).any(axis=1))
            elif name == "sub_mv":
                second = (b["ref_frame"][:, 1] > 0)
                ne = (x[:, :, 0] != y[:, :, 0]).any(axis=(1, 2)) | (second & (x[:, :, 1] != y[:, :, 1]).any(axis=(1, 2)))
            else:
                ne = x != y
                if ne.ndim > 1:
                    ne = ne.reshape(len(ne), -1).any(axis=1)
            if mask is not None:
                ne = ne & mask
            if ne.any():
                i = int(np.nonzero(ne)[0][0])
                return (f"frame {f}, block {i} (mi {int(b['mi_row'][i])},{int(b['mi_col'][i])} size {int(b['sb_type'][i])}): {name} "
                        f"{a[name][i].tolist()} != reference {b[name][i].tolist()}; mine {a[i]}, reference {b[i]}")
    return None


if __name__ == "__main__":
    import __graft_entry__ as g
    hip = g.load_pkg()
    for path in sys.argv[1:]:
        mine = parse_stream(hip, path)
        ref = reference_blocks(path)
        d = compare(mine, ref)
        print(os.path.basename(path), "OK: %d frames, %d blocks" % (len(mine), sum(len(x) for x in mine)) if d is None else d)
