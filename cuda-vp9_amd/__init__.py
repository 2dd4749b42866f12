"""cuda-vp9_amd — host-side Python mirror of the libvp9hip.so C-ABI (include/vp9hip.h).

This package is plumbing for tests, the benchmark and the multi-GPU batch harness: it loads
the in-tree HIP library with ctypes, mirrors the C structs as numpy dtypes and wraps the
entry points.  It contains no arithmetic and no CPU fallback: if libvp9hip.so is missing or
no HIP device is usable, calls raise.

The directory name carries a hyphen (the project's name); import it through
``__graft_entry__.load_pkg()`` or ``tests/vp9ref.load_hip()``, which register it as
``cuda_vp9_amd``.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libvp9hip.so")

# Several decoders in one process: every decoder has its own streams, and the ROCm runtime multiplexes a process's streams
# onto GPU_MAX_HW_QUEUES hardware queues (default 4) — kernels of two decoders that share a queue run one after the
# other.  Eight queues let eight decoders' launches overlap (tools/multi_stream_probe.py: 8 decoders 5.0 k -> 7.3 k
# frames/s).  The runtime reads the variable when it initialises, i.e. at the first HIP call of the process; a value the
# caller has set is kept.  (C hosts: export it before the process starts)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

_lib = None


class Vp9HipError(RuntimeError):
    pass


# ---- struct mirrors (include/vp9hip.h) ------------------------------------------------------
class Frame(ctypes.Structure):
    _fields_ = [("plane", ctypes.c_void_p * 3), ("stride", ctypes.c_int32 * 3),
                ("width", ctypes.c_int32 * 3), ("height", ctypes.c_int32 * 3),
                ("awidth", ctypes.c_int32 * 3), ("aheight", ctypes.c_int32 * 3),
                ("bit_depth", ctypes.c_int32), ("hbd", ctypes.c_int32)]


TXB_DTYPE = np.dtype([("coeff_off", "<u4"), ("x", "<u2"), ("y", "<u2"), ("plane", "u1"),
                      ("tx_size", "u1"), ("tx_type", "u1"), ("reserved", "u1"), ("eob", "<u2"),
                      ("reserved2", "<u2")])
INTER_DTYPE = np.dtype([("dst_x", "<i2"), ("dst_y", "<i2"), ("w", "u1"), ("h", "u1"),
                        ("plane", "u1"), ("flags", "u1"), ("pos_x", "<i4", (2,)),
                        ("pos_y", "<i4", (2,)), ("ref", "u1", (2,)), ("step_x", "u1", (2,)),
                        ("step_y", "u1", (2,)), ("reserved", "u1", (2,))])
INTRA_DTYPE = np.dtype([("coeff_off", "<u4"), ("x", "<u2"), ("y", "<u2"), ("plane", "u1"),
                        ("tx_size", "u1"), ("tx_type", "u1"), ("mode", "u1"), ("eob", "<u2"),
                        ("flags", "u1"), ("reserved", "u1")])
LFM_DTYPE = np.dtype([("left_y", "<u8", (4,)), ("above_y", "<u8", (4,)), ("int_4x4_y", "<u8"),
                      ("left_uv", "<u2", (4,)), ("above_uv", "<u2", (4,)), ("int_4x4_uv", "<u2"),
                      ("lfl_y", "u1", (64,)), ("reserved", "u1", (6,))])
ISLAND_DTYPE = np.dtype([("task_start", "<u4"), ("wave_off_start", "<u4"), ("n_waves", "<u4"), ("reserved", "<u4")])
assert TXB_DTYPE.itemsize == 16 and INTER_DTYPE.itemsize == 32 and ISLAND_DTYPE.itemsize == 16
assert INTRA_DTYPE.itemsize == 16 and LFM_DTYPE.itemsize == 160


# ---- host packers (include/vp9hip_pack.h) ---------------------------------------------------
BLOCK_DTYPE = np.dtype([("mi_row", "<i2"), ("mi_col", "<i2"), ("sb_type", "u1"), ("tx_size", "u1"), ("skip", "u1"),
                        ("interp_filter", "u1"), ("ref_frame", "i1", (2,)), ("mode", "u1"), ("uv_mode", "u1"),
                        ("sub_mode", "u1", (4,)), ("filter_level", "u1"), ("reserved", "u1", (3,)),
                        ("mv", "<i2", (2, 2)), ("sub_mv", "<i2", (4, 2, 2)), ("reserved2", "u1", (4,))])
assert BLOCK_DTYPE.itemsize == 64


class FrameParams(ctypes.Structure):
    _fields_ = [("width", ctypes.c_int32), ("height", ctypes.c_int32), ("ss_x", ctypes.c_int32),
                ("ss_y", ctypes.c_int32), ("bit_depth", ctypes.c_int32), ("hbd", ctypes.c_int32),
                ("lossless", ctypes.c_int32), ("log2_tile_cols", ctypes.c_int32),
                ("ref_width", ctypes.c_int32 * 3), ("ref_height", ctypes.c_int32 * 3),
                ("build_lf_masks", ctypes.c_int32), ("assume_coded", ctypes.c_int32),
                ("reserved", ctypes.c_int32 * 2)]


class CoeffLayout(ctypes.Structure):
    _fields_ = [("eob", ctypes.c_void_p * 3), ("eob_stride", ctypes.c_int32 * 3), ("eob_shift", ctypes.c_int32),
                ("block_off", ctypes.c_void_p), ("plane_base", ctypes.c_int64 * 3), ("total", ctypes.c_int64),
                ("regions", ctypes.c_void_p), ("n_regions", ctypes.c_int64),
                ("compact", ctypes.c_int32), ("reserved", ctypes.c_int32)]


class CoeffRegion(ctypes.Structure):
    _fields_ = [("plane", ctypes.c_int32), ("reserved", ctypes.c_int32), ("start", ctypes.c_int64), ("count", ctypes.c_int64)]


def _apply_tile_layout(cl, tile_layout, keep):
    """tile_layout: dict(block_off=uint32[n_blocks, 3], plane_base=[3], total=int, regions=[(plane, start, count)],
    compact=bool) — vp9hip_coeff_layout's optional part (slots placed by the caller, include/vp9hip_pack.h)."""
    bo = np.ascontiguousarray(tile_layout["block_off"], np.uint32)
    keep.append(bo)
    cl.block_off = bo.ctypes.data
    for p in range(3):
        cl.plane_base[p] = int(tile_layout["plane_base"][p])
    cl.total = int(tile_layout["total"])
    regs = tile_layout.get("regions") or []
    arr = (CoeffRegion * max(1, len(regs)))()
    for i, (pl, st, cnt) in enumerate(regs):
        arr[i].plane, arr[i].start, arr[i].count = int(pl), int(st), int(cnt)
    keep.append(arr)
    cl.regions = ctypes.addressof(arr)
    cl.n_regions = len(regs)
    cl.compact = int(bool(tile_layout.get("compact")))


INTER_CLASSES = 14  # VP9HIP_INTER_CLASSES
INTER_SHAPES = ((4, 4), (4, 8), (8, 4), (8, 8), (8, 16), (16, 8), (16, 16), (16, 32), (32, 16), (32, 32), (32, 64), (64, 32), (64, 64))


class Packed(ctypes.Structure):
    _fields_ = [("inter", ctypes.c_void_p), ("n_inter", ctypes.c_int32), ("inter_class_count", ctypes.c_int32 * INTER_CLASSES),
                ("txb", ctypes.c_void_p), ("n_txb", ctypes.c_int32), ("txb_size_count", ctypes.c_int32 * 4),
                ("intra_island_tasks", ctypes.c_void_p), ("n_intra_island_tasks", ctypes.c_int32),
                ("islands", ctypes.c_void_p), ("n_islands", ctypes.c_int32), ("n_islands_lds", ctypes.c_int32),
                ("island_wave_off", ctypes.c_void_p), ("n_island_wave_off", ctypes.c_int32),
                ("island_sb_expected", ctypes.c_void_p), ("island_row_pos", ctypes.c_void_p),
                ("intra_big_tasks", ctypes.c_void_p), ("n_intra_big_tasks", ctypes.c_int32),
                ("big_wave_start", ctypes.c_void_p), ("n_big_waves", ctypes.c_int32),
                ("intra_decode_order", ctypes.c_void_p), ("n_intra", ctypes.c_int32),
                ("n_intra_waves", ctypes.c_int32),
                ("coeff_base", ctypes.c_int64 * 3), ("coeff_count", ctypes.c_int64 * 3), ("coeff_total", ctypes.c_int64),
                ("lfm", ctypes.c_void_p), ("sb_rows", ctypes.c_int32), ("sb_cols", ctypes.c_int32),
                ("refs_used", ctypes.c_uint32)]


class HostFrame(ctypes.Structure):
    _fields_ = [("plane", ctypes.c_void_p * 3), ("stride", ctypes.c_int32 * 3), ("width", ctypes.c_int32),
                ("height", ctypes.c_int32), ("ss_x", ctypes.c_int32), ("ss_y", ctypes.c_int32),
                ("bit_depth", ctypes.c_int32), ("hbd", ctypes.c_int32)]


PHASE_INTER, PHASE_INTRA, PHASE_LF, PHASE_INTER_PRED, PHASE_INTER_RESID = 1, 2, 4, 8, 16


class LfThresh(ctypes.Structure):
    _fields_ = [("mblim", ctypes.c_uint8 * 64), ("lim", ctypes.c_uint8 * 64),
                ("hev_thr", ctypes.c_uint8 * 64)]


def lib():
    """The C-ABI library.  Raises if it has not been built — there is no fallback."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise Vp9HipError(
                f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
        L = ctypes.CDLL(LIB_PATH)
        vp = ctypes.c_void_p
        L.vp9hip_create.argtypes = [ctypes.c_int, ctypes.POINTER(vp)]
        L.vp9hip_destroy.argtypes = [vp]
        L.vp9hip_destroy.restype = None
        L.vp9hip_last_error.argtypes = [vp]
        L.vp9hip_last_error.restype = ctypes.c_char_p
        L.vp9hip_stream.argtypes = [vp]
        L.vp9hip_stream.restype = vp
        L.vp9hip_sync.argtypes = [vp]
        L.vp9hip_malloc.argtypes = [vp, ctypes.c_size_t]
        L.vp9hip_malloc.restype = vp
        L.vp9hip_free.argtypes = [vp, vp]
        L.vp9hip_free.restype = None
        L.vp9hip_memcpy_h2d.argtypes = [vp, vp, vp, ctypes.c_size_t]
        L.vp9hip_memcpy_d2h.argtypes = [vp, vp, vp, ctypes.c_size_t]
        L.vp9hip_memset.argtypes = [vp, vp, ctypes.c_int, ctypes.c_size_t]
        L.vp9hip_timer_begin.argtypes = [vp, ctypes.c_int]
        L.vp9hip_timer_end.argtypes = [vp, ctypes.c_int]
        L.vp9hip_timer_read.argtypes = [vp, ctypes.c_int, ctypes.POINTER(ctypes.c_float)]
        L.vp9hip_packer_create.argtypes = [ctypes.POINTER(vp)]
        L.vp9hip_packer_destroy.argtypes = [vp]
        L.vp9hip_packer_destroy.restype = None
        L.vp9hip_packer_error.argtypes = [vp]
        L.vp9hip_packer_error.restype = ctypes.c_char_p
        L.vp9hip_pack_frame.argtypes = [vp, ctypes.POINTER(FrameParams), vp, ctypes.c_int,
                                        ctypes.POINTER(CoeffLayout), ctypes.POINTER(Packed)]
        _lib = L
    return _lib


def _arr(ptr, n, dtype):
    """Copy n records of dtype out of packer-owned memory."""
    if not ptr or n <= 0:
        return np.zeros(0, dtype)
    buf = (ctypes.c_char * (n * np.dtype(dtype).itemsize)).from_address(ptr)
    return np.frombuffer(buf, dtype=dtype, count=n).copy()


class FeFrame(ctypes.Structure):
    """vp9hip_fe_frame (include/vp9hip_fe.h)."""
    _fields_ = [("show_existing", ctypes.c_int32), ("show_slot", ctypes.c_int32), ("show_frame", ctypes.c_int32),
                ("key_frame", ctypes.c_int32), ("intra_only", ctypes.c_int32), ("error_resilient", ctypes.c_int32),
                ("new_slot", ctypes.c_int32), ("ref_slot", ctypes.c_int32 * 3), ("refresh_flags", ctypes.c_int32),
                ("filter_level", ctypes.c_int32), ("sharpness", ctypes.c_int32), ("lf_thresh", LfThresh),
                ("params", FrameParams), ("blocks", ctypes.c_void_p), ("n_blocks", ctypes.c_int32),
                ("layout", CoeffLayout), ("dqcoeff", ctypes.c_void_p * 3), ("coeff_count", ctypes.c_int64),
                ("tile_cols", ctypes.c_int32), ("tile_rows", ctypes.c_int32)]


def ivf_packets(path):
    """The packets of an IVF file (libvpx/ivfdec.c: 32-byte file header, 12-byte frame headers)."""
    data = open(path, "rb").read()
    if data[:4] != b"DKIF":
        raise Vp9HipError(f"{path} is not an IVF file")
    pos = int.from_bytes(data[6:8], "little")
    while pos + 12 <= len(data):
        n = int.from_bytes(data[pos:pos + 4], "little")
        pos += 12
        yield data[pos:pos + n]
        pos += n


class FrontEnd:
    """vp9hip_fe (include/vp9hip_fe.h): the bitstream front-end, CPU only.  decoder: a Decoder whose page-locked
    memory the coefficient arrays come from (None: ordinary memory)."""

    def __init__(self, threads=0, decoder=None):
        L = lib()
        vp = ctypes.c_void_p
        L.vp9hip_fe_create.argtypes = [ctypes.POINTER(vp), vp, vp, vp, ctypes.c_int]
        L.vp9hip_fe_parse.argtypes = [vp, ctypes.c_char_p, ctypes.c_size_t, ctypes.POINTER(FeFrame)]
        L.vp9hip_fe_error.restype = ctypes.c_char_p
        L.vp9hip_fe_error.argtypes = [vp]
        L.vp9hip_fe_destroy.argtypes = [vp]
        L.vp9hip_fe_destroy.restype = None
        L.vp9hip_fe_split_superframe.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_uint32 * 8)]
        self.handle = vp()
        self._cb = None
        alloc = release = user = None
        if decoder is not None:
            A = ctypes.CFUNCTYPE(vp, vp, ctypes.c_size_t)
            R = ctypes.CFUNCTYPE(None, vp, vp)
            self._cb = (A(lambda u, n: L.vp9hip_decoder_host_alloc(decoder.handle, n)),
                        R(lambda u, p: L.vp9hip_decoder_host_free(decoder.handle, p)))
            alloc, release = ctypes.cast(self._cb[0], vp), ctypes.cast(self._cb[1], vp)
        if L.vp9hip_fe_create(ctypes.byref(self.handle), alloc, release, user, threads) != 0:
            raise Vp9HipError("vp9hip_fe_create failed")

    def frames_of(self, packet):
        """The frames a packet holds (superframe index), each as bytes."""
        sizes = (ctypes.c_uint32 * 8)()
        n = lib().vp9hip_fe_split_superframe(packet, len(packet), ctypes.byref(sizes))
        off = 0
        for k in range(n):
            if not (n > 1 and sizes[k] == 0):
                yield packet[off:off + sizes[k]]
            off += sizes[k]

    def parse(self, frame_bytes):
        fr = FeFrame()
        if lib().vp9hip_fe_parse(self.handle, frame_bytes, len(frame_bytes), ctypes.byref(fr)) != 0:
            raise Vp9HipError("vp9hip_fe_parse: " + lib().vp9hip_fe_error(self.handle).decode())
        return fr

    def close(self):
        if self.handle:
            lib().vp9hip_fe_destroy(self.handle)
            self.handle = ctypes.c_void_p()


class Decoder:
    """vp9hip_decoder (include/vp9hip_decoder.h): frame pool + transfers + phase sequencing."""

    def __init__(self, device=0):
        L = lib()
        vp = ctypes.c_void_p
        L.vp9hip_decoder_create.argtypes = [ctypes.c_int, ctypes.POINTER(vp)]
        L.vp9hip_decoder_destroy.argtypes = [vp]
        L.vp9hip_decoder_destroy.restype = None
        L.vp9hip_decoder_error.argtypes = [vp]
        L.vp9hip_decoder_error.restype = ctypes.c_char_p
        L.vp9hip_decoder_upload.argtypes = [vp, ctypes.c_int, ctypes.POINTER(HostFrame)]
        L.vp9hip_decoder_download.argtypes = [vp, ctypes.c_int, ctypes.POINTER(HostFrame)]
        L.vp9hip_decoder_alloc_slot.argtypes = [vp] + [ctypes.c_int] * 7
        L.vp9hip_decoder_begin_frame.argtypes = [vp, ctypes.POINTER(FrameParams), vp, ctypes.c_int,
                                                 ctypes.POINTER(CoeffLayout), ctypes.POINTER(vp * 3)]
        L.vp9hip_decoder_begin_frame_ex.argtypes = L.vp9hip_decoder_begin_frame.argtypes + [ctypes.c_int]
        L.vp9hip_decoder_set_timing.argtypes = [vp, ctypes.c_int]
        L.vp9hip_decoder_current_set.argtypes = [vp]
        L.vp9hip_decoder_select_set.argtypes = [vp, ctypes.c_int]
        L.vp9hip_decoder_host_alloc.argtypes = [vp, ctypes.c_size_t]
        L.vp9hip_decoder_host_alloc.restype = vp
        L.vp9hip_decoder_host_free.argtypes = [vp, vp]
        L.vp9hip_decoder_host_free.restype = None
        L.vp9hip_decoder_run.argtypes = [vp, ctypes.c_int, ctypes.POINTER(ctypes.c_int * 3), ctypes.c_int, vp, vp]
        L.vp9hip_decoder_sync.argtypes = [vp]
        L.vp9hip_decoder_last_run_ms.argtypes = [vp, ctypes.POINTER(ctypes.c_float)]
        self.handle = vp()
        rc = L.vp9hip_decoder_create(device, ctypes.byref(self.handle))
        if rc != 0:
            raise Vp9HipError(f"vp9hip_decoder_create failed ({rc}): {L.vp9hip_last_error(None).decode()}")

    def check(self, rc):
        if rc != 0:
            raise Vp9HipError(f"vp9hip_decoder call failed ({rc}): {lib().vp9hip_decoder_error(self.handle).decode()}")

    @staticmethod
    def host_frame(planes, width, height, bit_depth):
        h = HostFrame()
        for p, a in enumerate(planes):
            assert a.flags["C_CONTIGUOUS"]
            h.plane[p], h.stride[p] = a.ctypes.data, a.shape[1]
        h.width, h.height, h.ss_x, h.ss_y, h.bit_depth, h.hbd = width, height, 1, 1, bit_depth, int(bit_depth > 8)
        return h

    def upload(self, slot, planes, width, height, bit_depth):
        h = self.host_frame(planes, width, height, bit_depth)
        self.check(lib().vp9hip_decoder_upload(self.handle, slot, ctypes.byref(h)))

    def download(self, slot, planes, width, height, bit_depth):
        h = self.host_frame(planes, width, height, bit_depth)
        self.check(lib().vp9hip_decoder_download(self.handle, slot, ctypes.byref(h)))

    def alloc_slot(self, slot, width, height, bit_depth, clear=True):
        self.check(lib().vp9hip_decoder_alloc_slot(self.handle, slot, width, height, 1, bit_depth, int(bit_depth > 8),
                                                   int(clear)))

    def host_array(self, n, dtype):
        """A page-locked numpy array (vp9hip_decoder_host_alloc); freed with the decoder."""
        nbytes = max(1, int(n)) * np.dtype(dtype).itemsize
        p = lib().vp9hip_decoder_host_alloc(self.handle, nbytes)
        if not p:
            raise Vp9HipError("vp9hip_decoder_host_alloc failed")
        self._pinned = getattr(self, "_pinned", [])
        self._pinned.append(p)
        return np.frombuffer((ctypes.c_char * nbytes).from_address(p), dtype=dtype, count=int(n))

    def begin_frame(self, params, blocks, eob_planes=None, coef_planes=None, persistent=False, tile_layout=None):
        """persistent: coef_planes are page-locked arrays (host_array) left alone until the frame was run
        and synchronised: nothing is copied synchronously (VP9HIP_BEGIN_HOST_PERSISTENT).  Returns the
        ring set the frame went to."""
        blocks = np.ascontiguousarray(blocks, BLOCK_DTYPE)
        self._keep = [blocks]
        cl, dq = None, None
        if eob_planes is not None:
            cl = CoeffLayout()
            for p in range(3):
                a = np.ascontiguousarray(eob_planes[p], np.int32)
                self._keep.append(a)
                cl.eob[p], cl.eob_stride[p] = a.ctypes.data, a.shape[1]
            if tile_layout is not None:
                _apply_tile_layout(cl, tile_layout, self._keep)
        if coef_planes is not None:
            arrs = [np.ascontiguousarray(c, np.int32) for c in coef_planes]
            self._keep += arrs
            dq = (ctypes.c_void_p * 3)(*[a.ctypes.data if len(a) else None for a in arrs])
        self.check(lib().vp9hip_decoder_begin_frame_ex(self.handle, ctypes.byref(params), blocks.ctypes.data, len(blocks),
                                                       ctypes.byref(cl) if cl is not None else None,
                                                       ctypes.byref(dq) if dq is not None else None, int(bool(persistent))))
        return lib().vp9hip_decoder_current_set(self.handle)

    def begin_parsed(self, fr, persistent=True):
        """A frame as the front-end parsed it (FeFrame): slot, pack, upload.  Returns the ring set."""
        P = fr.params
        self.check(lib().vp9hip_decoder_alloc_slot(self.handle, fr.new_slot, P.width, P.height, P.ss_x, P.bit_depth, P.hbd, 0))
        dq = (ctypes.c_void_p * 3)(*fr.dqcoeff)
        self.check(lib().vp9hip_decoder_begin_frame_ex(self.handle, ctypes.byref(fr.params), fr.blocks, fr.n_blocks,
                                                       ctypes.byref(fr.layout), ctypes.byref(dq), int(bool(persistent))))
        return lib().vp9hip_decoder_current_set(self.handle)

    def run_parsed(self, fr):
        """All phases of a parsed frame (inter unless it is an intra frame, intra, loop filter when the frame has one)."""
        phases = PHASE_INTRA | (0 if (fr.key_frame or fr.intra_only) else PHASE_INTER) | (PHASE_LF if fr.filter_level else 0)
        rs = (ctypes.c_int * 3)(*fr.ref_slot)
        self.check(lib().vp9hip_decoder_run(self.handle, phases, ctypes.byref(rs), fr.new_slot, None,
                                            ctypes.addressof(fr.lf_thresh) if fr.filter_level else None))

    def set_timing(self, on):
        self.check(lib().vp9hip_decoder_set_timing(self.handle, int(bool(on))))

    def select_set(self, ring_set):
        self.check(lib().vp9hip_decoder_select_set(self.handle, ring_set))

    def run(self, phases, ref_slots, dst_slot, lfm=None, thresh=None):
        rs = (ctypes.c_int * 3)(*ref_slots)
        self.check(lib().vp9hip_decoder_run(self.handle, phases, ctypes.byref(rs), dst_slot,
                                            lfm.ctypes.data if lfm is not None else None,
                                            ctypes.addressof(thresh) if thresh is not None else None))

    def sync(self):
        self.check(lib().vp9hip_decoder_sync(self.handle))

    def last_run_ms(self):
        ms = ctypes.c_float()
        self.check(lib().vp9hip_decoder_last_run_ms(self.handle, ctypes.byref(ms)))
        return ms.value

    def close(self):
        if self.handle:
            for p in getattr(self, "_pinned", []):
                lib().vp9hip_decoder_host_free(self.handle, p)
            self._pinned = []
            lib().vp9hip_decoder_destroy(self.handle)
            self.handle = None


class Packer:
    """vp9hip_packer: decoded blocks -> work lists (host C, cuda-vp9_amd/csrc/vp9hip_pack.c)."""

    def __init__(self):
        self.handle = ctypes.c_void_p()
        if lib().vp9hip_packer_create(ctypes.byref(self.handle)) != 0:
            raise Vp9HipError("vp9hip_packer_create failed")

    def pack_only(self, params: FrameParams, blocks, eob_planes=None):
        """vp9hip_pack_frame without copying the lists out (host packing time measurements)."""
        return self.pack(params, blocks, eob_planes, copy=False)

    def pack(self, params: FrameParams, blocks, eob_planes=None, copy=True, tile_layout=None):
        """blocks: BLOCK_DTYPE array in decode order; eob_planes: three int32 2-D arrays indexed [y, x]
        (the reference's plane_eob layout) or None.  Returns a dict of numpy copies of the lists."""
        blocks = np.ascontiguousarray(blocks, BLOCK_DTYPE)
        cl = None
        keep = []
        if eob_planes is not None:
            cl = CoeffLayout()
            for p in range(3):
                a = np.ascontiguousarray(eob_planes[p], np.int32)
                keep.append(a)
                cl.eob[p] = a.ctypes.data
                cl.eob_stride[p] = a.shape[1]
            if tile_layout is not None:
                _apply_tile_layout(cl, tile_layout, keep)
        out = Packed()
        rc = lib().vp9hip_pack_frame(self.handle, ctypes.byref(params), blocks.ctypes.data, len(blocks),
                                     ctypes.byref(cl) if cl is not None else None, ctypes.byref(out))
        if rc != 0:
            raise Vp9HipError(f"vp9hip_pack_frame failed ({rc}): {lib().vp9hip_packer_error(self.handle).decode()}")
        n_sb = out.sb_rows * out.sb_cols
        if not copy:
            return out.n_inter
        return dict(
            inter_tasks=_arr(out.inter, out.n_inter, INTER_DTYPE), inter_class_count=list(out.inter_class_count),
            txb=_arr(out.txb, out.n_txb, TXB_DTYPE), txb_counts=list(out.txb_size_count),
            intra_island_tasks=_arr(out.intra_island_tasks, out.n_intra_island_tasks, INTRA_DTYPE),
            intra_islands=_arr(out.islands, out.n_islands, ISLAND_DTYPE),
            intra_island_wave_off=_arr(out.island_wave_off, out.n_island_wave_off, np.int32),
            island_sb_expected=_arr(out.island_sb_expected, out.sb_rows * out.sb_cols, np.int32),
            n_islands_lds=out.n_islands_lds,
            island_row_pos=_arr(out.island_row_pos, out.sb_rows if out.island_row_pos else 0, np.int32),
            intra_big_tasks=_arr(out.intra_big_tasks, out.n_intra_big_tasks, INTRA_DTYPE),
            intra_big_wave_start=_arr(out.big_wave_start, out.n_big_waves + 1, np.int32),
            intra_decode_order=_arr(out.intra_decode_order, out.n_intra, INTRA_DTYPE),
            n_waves=out.n_intra_waves, coeff_base=list(out.coeff_base), coeff_count=list(out.coeff_count),
            coeff_total=out.coeff_total, lfm=_arr(out.lfm, n_sb if out.lfm else 0, LFM_DTYPE),
            sb_rows=out.sb_rows, sb_cols=out.sb_cols, refs_used=out.refs_used)

    def close(self):
        if self.handle:
            lib().vp9hip_packer_destroy(self.handle)
            self.handle = None


class DevBuf:
    """A device allocation owned by a Context."""

    def __init__(self, ctx, nbytes):
        self.ctx, self.nbytes = ctx, int(nbytes)
        self.ptr = lib().vp9hip_malloc(ctx.handle, self.nbytes)
        if not self.ptr:
            raise Vp9HipError(ctx.error())

    def upload(self, arr):
        arr = np.ascontiguousarray(arr)
        assert arr.nbytes <= self.nbytes
        self.ctx.check(lib().vp9hip_memcpy_h2d(self.ctx.handle, self.ptr, arr.ctypes.data, arr.nbytes))
        return self

    def download(self, dtype, shape):
        out = np.empty(shape, dtype)
        assert out.nbytes <= self.nbytes
        self.ctx.check(lib().vp9hip_memcpy_d2h(self.ctx.handle, out.ctypes.data, self.ptr, out.nbytes))
        return out

    def free(self):
        if self.ptr:
            lib().vp9hip_free(self.ctx.handle, self.ptr)
            self.ptr = None

    def offset(self, nbytes):
        """A view `nbytes` into this allocation (not owning: never freed)."""
        return _DevView(self.ptr + int(nbytes))


class _DevView:
    def __init__(self, ptr):
        self.ptr = ptr


class DevFrame:
    """Three planes in HBM + the Frame descriptor the kernels take."""

    def __init__(self, ctx, width, height, bit_depth=8, hbd=None, ss_x=1, ss_y=1, stride_align=64):
        self.ctx = ctx
        self.bit_depth = bit_depth
        self.hbd = (bit_depth > 8) if hbd is None else bool(hbd)
        self.dtype = np.uint16 if self.hbd else np.uint8
        aw, ah = (width + 7) & ~7, (height + 7) & ~7
        self.dims = []
        for p in range(3):
            sx, sy = (ss_x, ss_y) if p else (0, 0)
            w, h = (width + sx) >> sx, (height + sy) >> sy
            paw, pah = aw >> sx, ah >> sy
            stride = (paw + stride_align - 1) // stride_align * stride_align
            self.dims.append((w, h, paw, pah, stride))
        self.bufs = [DevBuf(ctx, d[4] * d[3] * np.dtype(self.dtype).itemsize) for d in self.dims]
        self.desc = Frame()
        for p, d in enumerate(self.dims):
            self.desc.plane[p] = self.bufs[p].ptr
            self.desc.width[p], self.desc.height[p] = d[0], d[1]
            self.desc.awidth[p], self.desc.aheight[p], self.desc.stride[p] = d[2], d[3], d[4]
        self.desc.bit_depth = bit_depth
        self.desc.hbd = int(self.hbd)

    def upload(self, planes):
        """planes: three 2-D arrays of the aligned plane size (aheight x awidth)."""
        for p, (arr, d) in enumerate(zip(planes, self.dims)):
            full = np.zeros((d[3], d[4]), self.dtype)
            full[:, :d[2]] = arr[:d[3], :d[2]]
            self.bufs[p].upload(full)

    def download(self):
        out = []
        for p, d in enumerate(self.dims):
            full = self.bufs[p].download(self.dtype, (d[3], d[4]))
            out.append(full[:, :d[2]].copy())
        return out

    def free(self):
        for b in self.bufs:
            b.free()


class Context:
    def __init__(self, device=0):
        h = ctypes.c_void_p()
        rc = lib().vp9hip_create(device, ctypes.byref(h))
        if rc != 0:
            raise Vp9HipError(f"vp9hip_create({device}) failed ({rc}): "
                              f"{lib().vp9hip_last_error(None).decode()}")
        self.handle = h

    def error(self):
        return lib().vp9hip_last_error(self.handle).decode()

    def check(self, rc):
        if rc != 0:
            raise Vp9HipError(f"vp9hip call failed ({rc}): {self.error()}")

    def sync(self):
        self.check(lib().vp9hip_sync(self.handle))

    def timer_begin(self, slot):
        self.check(lib().vp9hip_timer_begin(self.handle, slot))

    def timer_end(self, slot):
        self.check(lib().vp9hip_timer_end(self.handle, slot))

    def timer_read(self, slot):
        ms = ctypes.c_float()
        self.check(lib().vp9hip_timer_read(self.handle, slot, ctypes.byref(ms)))
        return ms.value

    def stream(self):
        return lib().vp9hip_stream(self.handle)

    def alloc(self, arr_or_bytes):
        if isinstance(arr_or_bytes, (int, np.integer)):
            return DevBuf(self, arr_or_bytes)
        arr = np.ascontiguousarray(arr_or_bytes)
        return DevBuf(self, max(arr.nbytes, 16)).upload(arr)

    # ---- batched entry points --------------------------------------------------------------
    def idct_add_batch(self, d_blocks, size_count, d_coeffs, frame):
        sc = (ctypes.c_int32 * 4)(*[int(v) for v in size_count])
        self.check(lib().vp9hip_idct_add_batch(self.handle, ctypes.c_void_p(d_blocks.ptr), sc,
                                               ctypes.c_void_p(d_coeffs.ptr), ctypes.byref(frame.desc)))

    def inter_pred_batch(self, d_tasks, class_count, refs, dst):
        arr = (Frame * len(refs))(*[r.desc for r in refs])
        cc = (ctypes.c_int32 * INTER_CLASSES)(*[int(v) for v in class_count])
        self.check(lib().vp9hip_inter_pred_batch(self.handle, ctypes.c_void_p(d_tasks.ptr), cc,
                                                 arr, len(refs), ctypes.byref(dst.desc)))

    def intra_pred_waves(self, d_tasks, wave_start, d_coeffs, frame):
        ws = np.ascontiguousarray(wave_start, np.int32)
        self.check(lib().vp9hip_intra_pred_waves(
            self.handle, ctypes.c_void_p(d_tasks.ptr), ws.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)),
            len(ws) - 1, ctypes.c_void_p(d_coeffs.ptr if d_coeffs is not None else None),
            ctypes.byref(frame.desc)))

    def intra_pred_islands(self, d_tasks, d_islands, n_islands, d_wave_off, d_coeffs, frame):
        self.check(lib().vp9hip_intra_pred_islands(
            self.handle, ctypes.c_void_p(d_tasks.ptr), ctypes.c_void_p(d_islands.ptr), int(n_islands),
            ctypes.c_void_p(d_wave_off.ptr), ctypes.c_void_p(d_coeffs.ptr if d_coeffs is not None else None),
            ctypes.byref(frame.desc)))

    def intra_islands_lf(self, d_tasks, d_islands, n_islands, d_wave_off, d_coeffs, d_sb_expected, row_pos, d_lfm, sb_rows,
                         sb_cols, thresh, frame, planes=3):
        """Island walk (in LDS) and loop filter as one launch (vp9hip_intra_islands_lf).  row_pos: host int32 array
        (sb_rows entries) or None = every island in front of the filter's rows."""
        rp = None
        if row_pos is not None:
            rp = np.ascontiguousarray(row_pos, np.int32)
            assert len(rp) == sb_rows
        self.check(lib().vp9hip_intra_islands_lf(
            self.handle, ctypes.c_void_p(d_tasks.ptr), ctypes.c_void_p(d_islands.ptr), int(n_islands),
            ctypes.c_void_p(d_wave_off.ptr), ctypes.c_void_p(d_coeffs.ptr if d_coeffs is not None else None),
            ctypes.c_void_p(d_sb_expected.ptr), ctypes.c_void_p(rp.ctypes.data if rp is not None else None),
            ctypes.c_void_p(d_lfm.ptr), int(sb_rows), int(sb_cols),
            ctypes.byref(thresh), ctypes.byref(frame.desc), int(planes)))

    def loop_filter_frame(self, d_lfm, sb_rows, sb_cols, thresh, frame, planes=3):
        self.check(lib().vp9hip_loop_filter_frame(self.handle, ctypes.c_void_p(d_lfm.ptr), int(sb_rows),
                                                  int(sb_cols), ctypes.byref(thresh), ctypes.byref(frame.desc),
                                                  int(planes)))

    def close(self):
        if self.handle:
            lib().vp9hip_destroy(self.handle)
            self.handle = None


def sort_inter_tasks(tasks, hbd):
    """Group inter tasks into the classes vp9hip_inter_pred_batch takes (vp9hip_inter_class, include/vp9hip.h)."""
    compound = (tasks["flags"] & 1).astype(bool)
    unscaled = (tasks["step_x"][:, 0] == 16) & (tasks["step_y"][:, 0] == 16) & \
               (~compound | ((tasks["step_x"][:, 1] == 16) & (tasks["step_y"][:, 1] == 16)))  # the second reference only counts when used
    cls = np.full(len(tasks), INTER_CLASSES - 1, np.int32)
    for k, (w, h) in enumerate(INTER_SHAPES):
        cls[unscaled & (tasks["w"] == w) & (tasks["h"] == h)] = k
    order = np.argsort(cls, kind="stable")
    return tasks[order], [int((cls == k).sum()) for k in range(INTER_CLASSES)]


def sort_txb_by_size(blocks):
    """Group transform-block records by tx_size as vp9hip_idct_add_batch requires."""
    order = np.argsort(blocks["tx_size"], kind="stable")
    out = blocks[order]
    counts = [int((out["tx_size"] == s).sum()) for s in range(4)]
    return out, counts
