"""Full-size parity of exactly what bench.py times (run with -m gpu): every 1920x1080 workload is staged through
the same code (jsplayer_amd.workloads.StagedWorkload: jsp_stage_batch + jsp_staged_decode), decoded by the kernels
the bench launches, and every frame left in HBM is compared with the CPU oracle's digest of that frame
(tests/golden/bench_digests.json, written by tests/golden/make_bench_digests.py).  The kernel list of each workload
is pinned too, so a change of launch plan cannot slip past these tests."""
import pytest

from jsplayer_amd import workloads as wl

pytestmark = pytest.mark.gpu

CASES = [
    # workload, kernels the staged decode must launch
    ("msvideo1_16_1080p_keyframes_m1", ["msv1_fused_kernel"]),           # 3 x 512 M1 key frames from raw stream bytes
    ("msvideo1_16_1080p_keyframes_m1_hostdesc", ["msv1_blocks_kernel"]),  # host-built descriptor table
    ("msvideo1_8_1080p_keyframes_m1", ["msv1_fused_kernel"]),
    ("msvideo1_16_1080p_keyframes_solid", ["msv1_fused_kernel"]),         # one-slot codes only: 8192 blocks per 16 KiB tile, two staging windows
    ("msvideo1_16_1080p_keyframes_eight", ["msv1_fused_kernel"]),         # nine-slot codes only
    ("msvideo1_16_1080p_inter70", ["msv1_fused_kernel", "msv1_blocks_temporal_kernel"]),   # 511 inter frames: descriptor form of the fused parse + one temporal launch
    ("screenpressor_v4_1080p_iframes", ["sp_iframe_tile_kernel"]),        # 256 key frames, wave-per-tile kernel
    ("screenpressor_v2_1080p_iframes", ["sp_iframe_tile_kernel"]),        # the same through the version-2 range decoder
    ("screenpressor_v4_1080p_pclip300", ["sp_pframe_group_kernel"]),      # 2 x 299 inter frames, group kernel
]


@pytest.mark.parametrize("name,kernels", CASES, ids=[c[0] for c in CASES])
def test_staged_workload_matches_oracle_digests(name, kernels):
    gold = wl.golden_digests(name, 0)
    assert gold is not None, "run tests/golden/make_bench_digests.py"
    # (a replay of the inter-frame batch re-parses on the GPU: the block tables are poisoned before each replay, so frames
    # that come out right were rebuilt from tables the replay itself wrote)
    options = {"msv1_scrub_tables": "1"} if name == "msvideo1_16_1080p_inter70" else None
    work = wl.StagedWorkload(name, wl.build_clips(name, 0), options=options)
    try:
        for k in kernels:
            assert k in work.kernels(), work.kernels()
        assert work.frames_per_step == sum(len(g) for g in gold) - (len(gold) if work.inter else 0)
        work.step()
        work.sync()
        assert work.mismatches(gold) == []
        # a replay (what the timed steps are) writes the same frames: the destinations are overwritten with 0xEE bytes first, so
        # frames that match afterwards were written by the replay itself
        work.step()
        work.scrub()
        work.step()
        work.sync()
        assert work.mismatches(gold) == []
        # ... by the kernels named above: no batch was re-run through the descriptor path behind a look-back time-out
        assert work.lookback_fallbacks() == 0
    finally:
        work.close()


# BASELINE.json configs[4] (8 streams, one per GPU): what ranks >= 1 of the driver's N-GPU run decode.  Their clips come from other
# seeds than rank 0's (workloads._seed), their digests from the oracle (make_bench_digests.py wrote ranks 0 - 7 of the three
# workloads of the default bench line).  Ranks 1 and 7 whole, rank 4 its first clip
# (that the eight ranks' streams differ is checked on the CPU: test_bench_digests.py).
DRIVER_LINE = ["msvideo1_16_1080p_keyframes_m1", "screenpressor_v4_1080p_iframes", "screenpressor_v4_1080p_pclip300"]


@pytest.mark.parametrize("rank", [1, 4, 7])
@pytest.mark.parametrize("name", DRIVER_LINE)
def test_other_ranks_streams_match_oracle_digests(name, rank):
    gold = wl.golden_digests(name, rank)
    assert gold is not None, "run tests/golden/make_bench_digests.py --ranks 8"
    clips = wl.build_clips(name, rank)
    if rank not in (1, 7):
        clips, gold = clips[:1], gold[:1]
    work = wl.StagedWorkload(name, clips)
    try:
        work.step()
        work.scrub()
        work.step()
        work.sync()
        assert work.mismatches(gold) == []
        assert work.lookback_fallbacks() == 0
    finally:
        work.close()
