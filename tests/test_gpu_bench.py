"""bench.py end to end on the GPU box: the default line's contract, and the N > 1 path with two ranks sharing the one
device over gloo (LINK_BENCH_BACKEND=gloo: the branch exists for exactly this; the real multi-GPU run uses RCCL)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None, timeout=400):
    env = dict(os.environ)
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, cwd=ROOT, env=env, capture_output=True,
                       text=True, timeout=timeout)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_two_ranks_over_gloo_on_one_device():
    """`python bench.py --gpus 2` without a launcher re-executes itself under torch.distributed.run (127.0.0.1 rendezvous),
    every rank times its own frames, rank 0 prints ONE line with n_gpus = 2 and one gathered summary row per rank; the
    rows equal what each rank's frame gives on its own (rank r runs frames with seed r * 64 + k)."""
    d = _run(["--gpus", "2", "--steps", "3", "--warmup", "2", "--no-cpu-baseline"],
             {"LINK_BENCH_BACKEND": "gloo", "MASTER_ADDR": "127.0.0.1"})
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["scaling"] == "weak"
    assert [r["rank"] for r in d["ranks"]] == [0, 1]
    assert all(r["voxels"] == 100000 and 43000 < r["blocks"] < 43700 for r in d["ranks"])
    assert d["ranks"][0]["checksum"] != d["ranks"][1]["checksum"]           # different frames per rank
    assert d["value"] > 0 and d["config"]["frames_in_flight_per_gpu"] == 3
    # a step = one batch of 24 frames per GPU (8 rounds of the 3 in flight): 48 frames per step over the two ranks; value = voxels of all timed frames / time
    assert d["frames_per_step"] == 48 and abs(d["us_per_frame"] * 24 - d["ms_per_step"] * 1e3) < 0.4
    assert abs(d["value"] - 2 * 100000 / (d["us_per_frame"] * 1e-6)) < 1e-3 * d["value"]
    # SURVEY.md 8e side figures of the N > 1 line: the collective really spans both ranks, the summary gather sits inside a
    # reported end-to-end time, the full-tensor gather ([N, C] rows of every rank, two-phase pattern) is timed on its own
    mg = d["multi_gpu"]
    assert mg["backend"] == "gloo" and mg["world_size"] == 2 and mg["all_reduce_of_ones"] == 2.0 and mg["collective_spans_all_ranks"]
    assert mg["end_to_end_ms"] >= d["ms_per_step"] * d["steps"] * 0.5 and mg["end_to_end_ms"] > 0
    assert mg["full_tensor_gather_ms"] > 0 and mg["full_tensor_gather_ok"] and mg["full_tensor_gather_bytes_per_rank"] == 100000 * 64 * 4
    single = _run(["--gpus", "1", "--steps", "3", "--warmup", "2", "--no-cpu-baseline"])
    assert single["multi_gpu"] is None
    assert "gpu_state" in single and isinstance(single["gpu_state"], dict)          # clock / power state (amdsmi) or why it is missing
    assert single["n_gpus"] == 1 and single["ranks"][0]["blocks"] == d["ranks"][0]["blocks"]
    assert single["frames_per_step"] == 24 and single["three_frame_step"]["us_per_frame"] > 0 and abs(single["value"] - 100000 / (single["us_per_frame"] * 1e-6)) < 1e-3 * single["value"]
    assert abs(single["roofline"]["whole_step"]["us"] - single["us_per_frame"]) < 0.02
    assert abs(single["ranks"][0]["checksum"] - d["ranks"][0]["checksum"]) <= 1e-6 * abs(single["ranks"][0]["checksum"]) + 1e-3
    chk = single["timed_configuration_check"]
    assert chk["frames"] == 3 and chk["max_rel_err_vs_single_frame_geometry"] < 2e-6
    assert single["roofline"]["bound"] == "hbm" and 0 < single["roofline"]["frac"] < 1
    # round 6: the roofline object describes the TIMED region; per-kernel figures come from the committed rocprofv3 files of the timed
    # geometry (profiles/timed_geometry.json) and reproduce from them by hand
    rl = single["roofline"]
    assert abs(rl["frac"] - rl["whole_step"]["frac"]) < 1e-9 and rl["single_frame_geometry"]["step_median_us"] > 0
    tg = rl["timed_geometry"]
    # two ways through the K steps are timed -- three plans on three streams, the batch entry point -- and the faster one is the headline
    paths = single["paths"]
    assert paths["headline"] in ("streams", "batch") and single["config"]["path"] == paths["headline"]
    assert abs(paths[paths["headline"]]["us_per_frame"] - single["us_per_frame"]) < 1e-9 and paths["streams"]["us_per_frame"] > 0
    if paths["batch"] is not None:
        assert paths["batch"]["bitwise_equal_to_the_streams_rows"] and single["us_per_frame"] <= paths["streams"]["us_per_frame"] + 1e-9
        assert d["paths"]["batch"] is None                                    # two ranks on one device over gloo: streams only
    first = "insert" if paths["headline"] == "batch" else "index"
    assert tg is not None and set(tg["kernels"]) == {first, "premix_modsum", "gather_demod"}
    for k, v in tg["kernels"].items():
        assert abs(v["frac"] - v["alg_bytes_per_launch"] / (v["avg_us"] * 1e-6) / 8e12) < 2e-4, k
        assert v["live_event_us_this_run"] > 0
    assert rl["kernel"] == tg["kernels"][tg["dominant"]]["rocprof_name"]
    fs = single["frame_streams"]                       # the three frame streams were tested pair by pair for hardware queues of their own
    assert len(fs["max_delay_us_to_earlier_kept_stream"]) == 3 and max(fs["max_delay_us_to_earlier_kept_stream"]) < 75.0
    assert [r["rank"] for r in single["cpu_affinity"]["ranks"]] == [0] and [r["rank"] for r in d["cpu_affinity"]["ranks"]] == [0, 1]
    # side measurement: the same frames through the batch entry point (one insert + two persistent role kernels per call)
    be = single["batch_entry_point"]
    assert "error" not in be and be["us_per_frame"] > 0 and be["bitwise_equal_to_timed_configuration"] and be["frames_per_call"] == 48 and len(be["placement_trials_us_per_frame"]) >= 1


def test_bench_half_rows_check_reads_what_the_step_wrote():
    """`--io f16`: the timed steps write fp16 rows into the plan's per-dtype buffer; the line's checks must be computed from
    THAT buffer (round 3 compared the stale fp32 buffer with itself).  Half tolerances: the single-frame geometry gives the
    same rows up to the fp16 rounding of the output (2^-10 relative to the row's scale), the oracle on the rounded inputs
    agrees to 6e-3 (max-norm)."""
    d = _run(["--io", "f16", "--steps", "3", "--warmup", "2"], timeout=600)
    chk = d["timed_configuration_check"]
    assert chk["frames"] == 3 and 0.0 <= chk["max_rel_err_vs_single_frame_geometry"] < 2e-3
    err = d["cpu_baseline"]["gpu_vs_oracle_max_rel_err"]
    assert err == err and 0.0 < err < 6e-3, err
    assert d["dtype"].startswith("f16") and "fp16" in d["config"]["contraction"]
    assert d["ms_per_step_event_median"] > 0 and d["batch_period_events"]["n"] >= 50


def test_bench_cfg4_two_ranks_checksums_match_one_rank_and_the_committed_ones():
    """BASELINE.json configs[3] (`--workload cfg4`: 8 S-kitti frames sharded by frame): two ranks over gloo on the one device
    give the same eight per-frame checksums as one rank -- the frames are independent, whichever rank runs them -- and both
    equal tests/golden/g_cfg4_checksums.json (recorded from a 1-GPU run; what an 8-GPU run's line is to be compared with).
    Seeds 0, 1 and 7 of these frames are checked against the oracle in tests/test_gpu_encoder.py."""
    one = _run(["--workload", "cfg4", "--gpus", "1", "--steps", "1", "--warmup", "1"], timeout=600)
    two = _run(["--workload", "cfg4", "--gpus", "2", "--steps", "1", "--warmup", "1"],
               {"LINK_BENCH_BACKEND": "gloo", "MASTER_ADDR": "127.0.0.1"}, timeout=600)
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2 and two["scaling"] == "strong"
    assert one["config"]["frames"] == two["config"]["frames"] == 8
    assert one["config"]["frames_per_rank"] == 8 and two["config"]["frames_per_rank"] == 4
    a, b = one["frame_checksums"], two["frame_checksums"]
    assert len(a) == len(b) == 8
    for x, y in zip(a, b):
        assert abs(x - y) <= 1e-6 * abs(x), (a, b)
    with open(os.path.join(ROOT, "tests", "golden", "g_cfg4_checksums.json")) as f:
        gold = json.load(f)
    assert gold["voxels"] == one["config"]["voxels"]
    for x, y in zip(a, gold["frame_checksums"]):
        assert abs(x - y) <= 1e-5 * abs(y), (a, gold["frame_checksums"])
