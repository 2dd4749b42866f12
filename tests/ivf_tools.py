"""IVF container helpers for the tests (libvpx/ivfdec.c, ivfenc.c: 32-byte file header, 12-byte frame headers)."""
import struct


def read_ivf(path):
    """(header bytes, [(pts, payload), ...])"""
    data = open(path, "rb").read()
    assert data[:4] == b"DKIF"
    hdr_len = struct.unpack_from("<H", data, 6)[0]
    frames, pos = [], hdr_len
    while pos + 12 <= len(data):
        size, pts = struct.unpack_from("<IQ", data, pos)
        pos += 12
        if pos + size > len(data):
            break
        frames.append((pts, data[pos:pos + size]))
        pos += size
    return data[:hdr_len], frames


def write_ivf(path, header, frames):
    hdr = bytearray(header)
    struct.pack_into("<I", hdr, 24, len(frames))
    with open(path, "wb") as f:
        f.write(hdr)
        for i, (_, payload) in enumerate(frames):
            f.write(struct.pack("<IQ", len(payload), i))
            f.write(payload)


def concat_ivf(paths, out):
    """The frames of several IVF files one after the other in one file (each starts with a key frame, so the result is
    a valid stream whose frame size changes in mid-stream; the file header is the first file's)."""
    header, frames = None, []
    for p in paths:
        h, fr = read_ivf(p)
        header = header or h
        frames += fr
    write_ivf(out, header, frames)
    return len(frames)
