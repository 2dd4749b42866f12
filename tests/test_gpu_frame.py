"""GPU parity on whole synthetic frames: all four kernel families chained in the reference's
phase order vs the sequential oracle; bit-exact planes and equal per-frame MD5."""
import numpy as np
import pytest

import frame_check

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("W,H,bd,kw", [
    (352, 288, 8, {}),
    (360, 200, 8, dict(intra_frac=0.5, sharpness=4)),
    (320, 192, 10, dict(compound_frac=0.5)),
    (256, 128, 8, dict(all_intra=True)),
    (1920, 1080, 8, {}),
    # BASELINE.json configs[3]: 2160p 8-bit all-inter, high motion (stresses the convolve)
    (3840, 2160, 8, dict(intra_frac=0.0, compound_frac=0.3, skip_frac=0.6)),
    # BASELINE.json configs[4]: 1080p 10-bit (highbd transform / convolve path)
    (1920, 1080, 10, {}),
])
def test_frame_pipeline_matches_oracle(hip, oracle, W, H, bd, kw):
    import workload
    import cuda_vp9_amd.pipeline as pipeline
    wl = workload.make_frame_workload(W, H, seed=W + H + bd, bd=bd, **kw)
    ctx = hip.Context(0)
    job = pipeline.FrameJob(ctx, wl)
    job.clear_dst()
    job.run()
    ctx.sync()
    got = job.download()
    expect, _ = frame_check.oracle_frame(oracle, wl)
    for p in range(3):
        bad = np.argwhere(got[p] != expect[p])
        assert bad.size == 0, f"plane {p}: {len(bad)} px differ, first {bad[:5]}"
    assert frame_check.frame_md5(got, wl) == frame_check.frame_md5(expect, wl)
    # idempotence of the device path: re-running on the same job gives the same frame
    job.clear_dst()
    job.run()
    ctx.sync()
    again = job.download()
    assert all(np.array_equal(a, b) for a, b in zip(got, again))
    # the sequential form (islands, then the loop filter) and the overlapped one agree
    job.overlap = False
    job.clear_dst()
    job.run()
    ctx.sync()
    seq = job.download()
    assert all(np.array_equal(a, b) for a, b in zip(got, seq))
    # ... and so does the fused launch with every island in front of the filter's rows (no h_row_pos)
    job.overlap, job.row_pos = True, False
    job.clear_dst()
    job.run()
    ctx.sync()
    first = job.download()
    assert all(np.array_equal(a, b) for a, b in zip(got, first))
    job.free()
    ctx.close()


def test_loop_filter_handoff_is_stable_under_repetition(hip, oracle):
    """The row-walking loop filter hands pixels between workgroups inside one launch; a stale
    read would show up as a run-to-run difference.  40 repetitions on a 1440p frame, each compared
    with the (deterministic) oracle result."""
    import workload
    import cuda_vp9_amd.pipeline as pipeline
    wl = workload.make_frame_workload(2560, 1440, seed=99)
    ctx = hip.Context(0)
    job = pipeline.FrameJob(ctx, wl)
    expect, _ = frame_check.oracle_frame(oracle, wl)
    want = frame_check.frame_md5(expect, wl)
    for it in range(40):
        job.run()
        ctx.sync()
        got = job.download()
        assert frame_check.frame_md5(got, wl) == want, f"iteration {it}"
    job.free()
    ctx.close()


def test_loop_filter_gives_up_cleanly_when_an_island_never_reports(hip, oracle):
    """The overlapped pair (island walk || loop filter): a filter row waits, bounded, for the islands around its
    superblock.  A gate that can never open (here: one more island announced for a superblock than exists)
    must end as an ERROR from vp9hip_sync — never as a hang or a silently wrong frame — stay reported
    exactly once, and leave the context usable."""
    import workload
    import cuda_vp9_amd.pipeline as pipeline
    wl = workload.make_frame_workload(352, 288, seed=5, intra_frac=0.3)
    ctx = hip.Context(0)
    job = pipeline.FrameJob(ctx, wl)
    job.run()
    ctx.sync()
    good = job.download()
    bad = wl["island_sb_expected"].copy()
    bad[len(bad) // 2] += 1
    good_buf, job.d_sb_expected = job.d_sb_expected, ctx.alloc(bad)
    job.run()
    # ... and name the wait at the ROOT of the chain (rows below the stalled one wait for rows, with twice the time):
    # the islands of that superblock, one mark short, every workgroup of the launch started
    with pytest.raises(hip.Vp9HipError, match="gave up waiting") as ei:
        ctx.sync()
    import re
    m = re.search(r"waited for the intra islands of a superblock \(had (\d+) of (\d+); (\d+) of the launch's (\d+) workgroups", str(ei.value))
    assert m, str(ei.value)
    assert int(m.group(1)) + 1 == int(m.group(2)) == int(bad[len(bad) // 2]) and m.group(3) == m.group(4), str(ei.value)
    ctx.sync()  # reported once
    job.d_sb_expected.free()
    job.d_sb_expected = good_buf
    job.run()
    ctx.sync()
    again = job.download()
    assert all(np.array_equal(a, b) for a, b in zip(good, again))
    job.free()
    ctx.close()
