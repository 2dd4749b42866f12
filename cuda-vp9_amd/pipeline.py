"""pipeline.py — one frame through the HIP reconstruction path, phases in the reference's
order (decode_tiles, vp9/decoder/vp9_decodeframe.c:2536-2620): inter prediction -> residual of
inter blocks -> wave-ordered intra prediction (+ residual) -> loop filter.  Device-resident:
work lists and reference frames are uploaded once; run() only enqueues kernels."""
import ctypes

import numpy as np

from . import Context, DevFrame, LfThresh


class FrameJob:
    def __init__(self, ctx: Context, wl: dict):
        self.ctx, self.wl = ctx, wl
        W, H, bd, hbd = wl["width"], wl["height"], wl["bd"], wl["hbd"]
        self.refs = []
        for planes in wl["refs"]:
            fr = DevFrame(ctx, W, H, bit_depth=bd, hbd=hbd)
            fr.upload(planes)
            self.refs.append(fr)
        self.dst = DevFrame(ctx, W, H, bit_depth=bd, hbd=hbd)
        from . import sort_inter_tasks
        self.inter_sorted, self.inter_counts = sort_inter_tasks(wl["inter_tasks"], hbd)
        self.d_inter = ctx.alloc(self.inter_sorted) if len(self.inter_sorted) else None
        self.d_txb = ctx.alloc(wl["txb"]) if len(wl["txb"]) else None
        self.d_coeffs = ctx.alloc(wl["coeffs"])
        self.d_intra = ctx.alloc(wl["intra_sorted"]) if len(wl["intra_sorted"]) else None
        self.d_isl_tasks = ctx.alloc(wl["intra_island_tasks"]) if len(wl["intra_island_tasks"]) else None
        self.d_islands = ctx.alloc(wl["intra_islands"]) if len(wl["intra_islands"]) else None
        self.d_isl_woff = ctx.alloc(wl["intra_island_wave_off"])
        self.d_big_tasks = ctx.alloc(wl["intra_big_tasks"]) if len(wl["intra_big_tasks"]) else None
        self.d_sb_expected = ctx.alloc(wl["island_sb_expected"])
        self.use_islands = True
        self.overlap = True
        self.row_pos = True  # False: every island in front of the filter's rows in the fused launch's grid
        self.d_lfm = ctx.alloc(wl["lfm"])
        self.th = LfThresh()
        mblim, lim, hev = wl["thresholds"]
        for i in range(64):
            self.th.mblim[i], self.th.lim[i], self.th.hev_thr[i] = int(mblim[i]), int(lim[i]), int(hev[i])

    def clear_dst(self):
        for b in self.dst.bufs:
            self.ctx.check(__import__("cuda_vp9_amd").lib().vp9hip_memset(self.ctx.handle, b.ptr, 0, b.nbytes))

    def run(self, phases=("inter", "txb", "intra", "lf")):
        wl, ctx = self.wl, self.ctx
        if "inter" in phases and self.d_inter is not None:
            ctx.inter_pred_batch(self.d_inter, self.inter_counts, self.refs, self.dst)
        if "txb" in phases and self.d_txb is not None:
            ctx.idct_add_batch(self.d_txb, wl["txb_counts"], self.d_coeffs, self.dst)
        # intra + loop filter of the same frame as one launch (islands walked in LDS where they fit, the filter's
        # rows gated on the islands they depend on) unless the frame has components too large for an island (key
        # frames) or only one of the two phases is asked for
        if ("intra" in phases and "lf" in phases and self.use_islands and self.overlap and self.d_islands is not None
                and self.d_big_tasks is None and wl["sb_rows"] <= 128 and wl["sb_cols"] <= 128):
            ctx.intra_islands_lf(self.d_isl_tasks, self.d_islands, len(wl["intra_islands"]), self.d_isl_woff,
                                 self.d_coeffs, self.d_sb_expected, wl["island_row_pos"] if self.row_pos else None,
                                 self.d_lfm, wl["sb_rows"], wl["sb_cols"], self.th, self.dst, 3)
            return
        if "intra" in phases and self.d_intra is not None:
            if self.use_islands:
                # islands: one launch; components too large for one workgroup keep per-wave launches
                if self.d_islands is not None:
                    ctx.intra_pred_islands(self.d_isl_tasks, self.d_islands, len(wl["intra_islands"]),
                                           self.d_isl_woff, self.d_coeffs, self.dst)
                if self.d_big_tasks is not None:
                    ctx.intra_pred_waves(self.d_big_tasks, wl["intra_big_wave_start"], self.d_coeffs, self.dst)
            else:
                ctx.intra_pred_waves(self.d_intra, wl["wave_start"], self.d_coeffs, self.dst)
        if "lf" in phases:
            ctx.loop_filter_frame(self.d_lfm, wl["sb_rows"], wl["sb_cols"], self.th, self.dst, 3)

    def download(self):
        return self.dst.download()

    def free(self):
        for fr in self.refs + [self.dst]:
            fr.free()
        for b in (self.d_inter, self.d_txb, self.d_coeffs, self.d_intra, self.d_lfm, self.d_isl_tasks, self.d_islands,
                  self.d_isl_woff, self.d_big_tasks, self.d_sb_expected):
            if b is not None:
                b.free()


def algorithmic_bytes(wl):
    """Algorithmic HBM bytes of one frame pass, per kernel family (SURVEY §8d formulas)."""
    bps = 2 if wl["hbd"] else 1
    it = wl["inter_tasks"]
    px = it["w"].astype(np.int64) * it["h"]
    nref = 1 + (it["flags"] & 1)
    conv = int((px * nref).sum() * bps + px.sum() * bps + 32 * len(it))
    tb = wl["txb"]
    n2 = (4 << tb["tx_size"].astype(np.int64)) ** 2
    dc = np.where(tb["tx_size"] == 0, tb["eob"] <= 1, tb["eob"] == 1)
    idct = int((np.where(dc, 1, n2) * 4).sum() + 2 * n2.sum() * bps + 16 * len(tb))
    ia = wl["intra_sorted"]
    bs = 4 << ia["tx_size"].astype(np.int64)
    intra = int((bs * bs * bps).sum() + ((3 * bs + 1) * bps).sum() + 16 * len(ia) + ((ia["eob"] > 0) * bs * bs * 4).sum())
    P = sum(aw * ah for (aw, ah) in wl["dims"])
    lf = int(2 * P * bps + 160 * wl["sb_rows"] * wl["sb_cols"])
    return dict(convolve=conv, idct_add=idct, intra=intra, loop_filter=lf)
