#!/usr/bin/env python3
"""The filtering wave's time per superblock step on a frame of a REAL stream (probe build, tools/build_probe_lib.sh):
    python tools/row_stamps_real.py [stream] [frame index]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import __graft_entry__ as g
hip = g.load_pkg()
hip.LIB_PATH = os.path.join(ROOT, "tools", "build", "libvp9hip_stamps.so")
name = sys.argv[1] if len(sys.argv) > 1 else "S-1440"
want = int(sys.argv[2]) if len(sys.argv) > 2 else 21
ivf = os.path.join(ROOT, "tests", "streams_big", name + ".ivf")
dec = hip.Decoder(0)
fe = hip.FrontEnd(threads=0, decoder=dec)
index = 0
done = False
for pkt in hip.ivf_packets(ivf):
    for data in fe.frames_of(pkt):
        fr = fe.parse(data)
        if fr.show_existing:
            continue
        ring = dec.begin_parsed(fr)
        dec.run_parsed(fr)
        dec.sync()
        if index == want:
            for _ in range(5):
                dec.select_set(ring)
                dec.run_parsed(fr)
            dec.sync()
            st = np.zeros((4096, 8), np.int64)
            assert hip.lib().vp9hip_debug_stamps(st.ctypes.data_as(ctypes.c_void_p), 4096) == 0
            rows = st[(st[:, 7] > 0) & (st[:, 2] > 0) & (st[:, 2] < 10_000_000)]  # (rows carry sums in 2..6, islands clock values)
            rows = rows[rows[:, 0] > rows[:, 0].max() - 200_000]  # the last launch only (slots past its grid are stale)
            sb_cols = (fr.params.width + 63) // 64
            print(f"{name} frame {index}: {fr.n_blocks} blocks, filter level {int(fr.filter_level)}, last run {dec.last_run_ms() * 1e3:.0f} us GPU; {len(rows)} filter row workgroups")
            t0 = rows[:, 0].min()
            a = rows[:, 2:7] / 100.0
            # luma rows have 40 steps, chroma rows 40 steps of 32 samples: per step of the row's own count
            print("per superblock step, us, mean over the rows (min..max): vertical pass | barrier | horizontal pass | barrier | (wave 1 beside the horizontal pass)")
            per = a / sb_cols
            for k, nm in enumerate(("vertical pass", "barrier behind it", "horizontal pass", "barrier behind it", "(wave 1's phase-B work)")):
                print(f"  {nm:26s} {per[:, k].mean():5.2f}  ({per[:, k].min():5.2f} .. {per[:, k].max():5.2f})")
            print(f"  rows end at {(rows[:, 7].max() - t0) / 100.0:.1f} us after the first row's start")
            order = np.argsort(rows[:, 0])
            print("  (start, first superblock ready, end, sum of the four per step) in start order, every third workgroup:")
            for i in order[::3]:
                print(f"    {(rows[i, 0] - t0) / 100.0:7.1f} {(rows[i, 1] - t0) / 100.0:7.1f} {(rows[i, 7] - t0) / 100.0:7.1f}   {per[i, :4].sum():5.2f}  = {per[i, 0]:.2f} + {per[i, 1]:.2f} + {per[i, 2]:.2f} + {per[i, 3]:.2f}")
            done = True
            break
        index += 1
    if done:
        break
fe.close(); dec.close()
