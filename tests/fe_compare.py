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


class FeFrame(ctypes.Structure):
    pass


def _fe_frame_struct(hip):
    class LfThresh(ctypes.Structure):
        _fields_ = [("mblim", ctypes.c_uint8 * 64), ("lim", ctypes.c_uint8 * 64), ("hev_thr", ctypes.c_uint8 * 64)]

    class F(ctypes.Structure):
        _fields_ = [("show_existing", ctypes.c_int32), ("show_slot", ctypes.c_int32), ("show_frame", ctypes.c_int32),
                    ("key_frame", ctypes.c_int32), ("intra_only", ctypes.c_int32), ("error_resilient", ctypes.c_int32),
                    ("new_slot", ctypes.c_int32), ("ref_slot", ctypes.c_int32 * 3), ("refresh_flags", ctypes.c_int32),
                    ("filter_level", ctypes.c_int32), ("sharpness", ctypes.c_int32), ("lf_thresh", LfThresh),
                    ("params", hip.FrameParams), ("blocks", ctypes.c_void_p), ("n_blocks", ctypes.c_int32),
                    ("layout", hip.CoeffLayout), ("dqcoeff", ctypes.c_void_p * 3), ("coeff_count", ctypes.c_int64),
                    ("tile_cols", ctypes.c_int32), ("tile_rows", ctypes.c_int32)]
    return F


def ivf_frames(path):
    data = open(path, "rb").read()
    assert data[:4] == b"DKIF"
    pos = int.from_bytes(data[6:8], "little")
    while pos + 12 <= len(data):
        n = int.from_bytes(data[pos:pos + 4], "little")
        pos += 12
        yield data[pos:pos + n]
        pos += n


def parse_stream(hip, path, threads=1):
    """Every decoded frame's block records as the product's front-end parses them."""
    lib = hip.lib()
    F = _fe_frame_struct(hip)
    lib.vp9hip_fe_create.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
    lib.vp9hip_fe_parse.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_size_t, ctypes.POINTER(F)]
    lib.vp9hip_fe_error.restype = ctypes.c_char_p
    lib.vp9hip_fe_error.argtypes = [ctypes.c_void_p]
    lib.vp9hip_fe_destroy.argtypes = [ctypes.c_void_p]
    lib.vp9hip_fe_split_superframe.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_uint32 * 8)]
    fe = ctypes.c_void_p()
    os.environ["VP9HIP_FE_CHECKSUMS"] = "1"  # read at creation: per-block checksums over eobs + coefficients
    assert lib.vp9hip_fe_create(ctypes.byref(fe), None, None, None, threads) == 0
    del os.environ["VP9HIP_FE_CHECKSUMS"]
    frames = []
    try:
        for pkt in ivf_frames(path):
            sizes = (ctypes.c_uint32 * 8)()
            nf = lib.vp9hip_fe_split_superframe(pkt, len(pkt), ctypes.byref(sizes))
            off = 0
            for k in range(nf):
                if nf > 1 and sizes[k] == 0:
                    continue
                fr = F()
                rc = lib.vp9hip_fe_parse(fe, pkt[off:off + sizes[k]], sizes[k], ctypes.byref(fr))
                off += sizes[k]
                if rc:
                    raise RuntimeError(f"frame {len(frames)}: vp9hip_fe_parse: {lib.vp9hip_fe_error(fe).decode()}")
                if fr.show_existing:
                    continue
                buf = (ctypes.c_char * (64 * fr.n_blocks)).from_address(fr.blocks)
                frames.append(np.frombuffer(bytes(buf), REC).copy())
    finally:
        lib.vp9hip_fe_destroy(fe)
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
