#!/usr/bin/env python3
"""PCIe-inclusive rate of the reference call surface: one 2560x1440 frame through
wrap_cuda_inter_prediction + wrap_cuda_intra_prediction (shim/build/libshimtest.so harness), host
buffers in, host frame out.  Prints the wrappers' own gpu_copy / gpu_run figures and wall time."""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import __graft_entry__ as g
hip = g.load_pkg()
import blockgen
import workload
W, H, bd = 2560, 1440, int(sys.argv[1]) if len(sys.argv) > 1 else 8
coefficient_mode = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(7)
dt = np.uint16 if bd > 8 else np.uint8
t0 = time.time()
blocks = blockgen.gen_blocks(rng, W, H, hip.BLOCK_DTYPE, intra_frac=0.08, skip_frac=0.35)
coef, eob = blockgen.gen_coeffs(rng, blocks, W, H, bd)
aw, ah = (W + 7) & ~7, (H + 7) & ~7
dims = [(aw, ah), (aw // 2, ah // 2), (aw // 2, ah // 2)]
refs = [[workload.smooth_noise(rng, d[1], d[0], bd, sigma=1.5).astype(dt) for d in dims] for _ in range(3)]
print(f"generated {len(blocks)} blocks, {sum(len(c) for c in coef)} coefficients in {time.time()-t0:.1f} s", flush=True)
lib = ctypes.CDLL(os.path.join(ROOT, "shim", "build", "libshimtest.so"))
lib.shimtest_create.restype = ctypes.c_void_p
h = lib.shimtest_create()
recs = blockgen.to_ref_records(blocks)
got = [np.zeros((d[1], d[0]), dt) for d in dims]
flat = [a for r in refs for a in r]
ref_ptrs = (ctypes.c_void_p * 9)(*[a.ctypes.data for a in flat])
rw = (ctypes.c_int * 3)(W, W, W); rh = (ctypes.c_int * 3)(H, H, H)
dq = (ctypes.c_void_p * 3)(*[c.ctypes.data for c in coef])
eobk = [np.ascontiguousarray(e) for e in eob]
eobp = (ctypes.c_void_p * 3)(*[e.ctypes.data for e in eobk])
res = None
if not coefficient_mode:
    res = [np.zeros((d[1], d[0]), np.int64) for d in dims]
resp = (ctypes.c_void_p * 3)(*[r.ctypes.data for r in res]) if res else None
outp = (ctypes.c_void_p * 3)(*[x.ctypes.data for x in got])
times = (ctypes.c_double * 4)(); err = ctypes.create_string_buffer(512)
gpu_lf = int(sys.argv[3]) if len(sys.argv) > 3 else 0
# opts: GPU loop filter, sharpness, new_fb_idx, LAST/GOLDEN/ALTREF buffer indices, fill mask, poison mask
opts = (ctypes.c_int32 * 8)(gpu_lf, 0, 3, 0, 1, 2, 7, 0)
for it in range(6):
    t0 = time.perf_counter()
    rc = lib.shimtest_frame(ctypes.c_void_p(h), recs.ctypes.data_as(ctypes.c_void_p), len(recs), W, H, bd, int(bd > 8), 2, 0, 1,
                            coefficient_mode, ref_ptrs, rw, rh, dq, eobp, resp, outp, times, err, 512, opts)
    wall = time.perf_counter() - t0
    assert rc == 0, err.value
    print(f"frame {it}: inter copy {times[0]*1e3:.2f} ms run {times[1]*1e3:.3f} ms | intra copy {times[2]*1e3:.2f} ms run {times[3]*1e3:.3f} ms"
          f" | wrappers total {(sum(times))*1e3:.2f} ms | harness wall {wall*1e3:.1f} ms", flush=True)
lib.shimtest_destroy(ctypes.c_void_p(h))
