"""bench.py's N > 1 path: one stream cut into contiguous time shards, each rank starting from the stream's exact
carrier phase at its shard, digests exchanged, no data-path collective.

  * on the CPU box: `bench.py --dry-run` under torch.distributed.run with the gloo backend, world sizes 1 and 2 — the
    sharding, the shard seeds (gpsbb_chain_carrier_host) and the digest exchange are the product's, the rendering is
    the CPU oracle's; the digest of the whole stream must not depend on the number of ranks;
  * on the GPU box (-m gpu): the real thing, two ranks sharing the one device (GPSBB_BENCH_BACKEND=gloo), through the
    device-only ring and the pinned gather, small blocks; the second rank's seed comes from the device-side chain."""
import json
import os
import socket
import subprocess
import sys

import pytest

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(world, extra, env_extra=None, timeout=600, expect_rc=0):
    env = dict(os.environ)
    env.update(env_extra or {})
    if world == 1:
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"] + extra
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
               "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
               os.path.join(ROOT, "bench.py"), "--gpus", str(world)] + extra
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout)
    assert out.returncode == expect_rc, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]      # ONE JSON line, from rank 0
    return json.loads(lines[0])


def test_dry_run_digest_does_not_depend_on_the_number_of_ranks(pkg):
    args = ["--dry-run", "--steps", "1", "--push-blocks", "1", "--nsamp", "20000", "--fs", "25e6", "--m1-push-blocks", "1", "--m1-nsamp", "9000"]
    one = _run(1, args)
    two = _run(2, args)
    assert one["blocks"] == two["blocks"] == 8
    assert one["n_ranks"] == 1 and two["n_ranks"] == 2
    assert one["stream_digest"] == two["stream_digest"]
    # the 2.6 MS/s stream (12 channels, the reference's geometry) is cut and seeded the same way, at every N
    assert one["m1"]["blocks"] == two["m1"]["blocks"] == 8 and one["m1"]["nch"] == 12 and one["m1"]["fs"] == 2.6e6
    assert one["m1"]["stream_digest"] == two["m1"]["stream_digest"] != one["stream_digest"]


@pytest.mark.gpu
def test_two_ranks_through_the_device_path(pkg):
    """One rank and two ranks (sharing the one device, gloo) render the same stream: the second rank's shard starts from
    the seed the device-side chain computes (gpsbb_chain_carrier over the first shard), and the digest of the
    end-of-block states of all blocks, gathered over the ranks, is the 1-rank value; blocks read back from the HBM-only
    ring equal the oracle's on every rank; the line carries the slowest rank's seed time, the seed-inclusive value and
    every block cross-checked and the node driver validated at either N; so are the 2.6 MS/s stream (12 ch, the reference's
    geometry: cut, seeded and cross-checked like the headline's), the CPU baselines and the drop-in call's latency; the resident
    legs at N = 1 only."""
    args = ["--steps", "2", "--warmup", "1", "--repeats", "2", "--push-blocks", "8", "--nsamp", "200000", "--depth", "3",
            "--cpu-budget", "0.4", "--parity-blocks", "2", "--m1-push-blocks", "40", "--fill-calls", "12"]
    env = {"GPSBB_BENCH_BACKEND": "gloo"}
    one = _run(1, args, env)
    two = _run(2, args, env)
    for r, n in ((one, 1), (two, 2)):
        assert r["n_gpus"] == n and r["scaling"] == "strong" and r["steps"] == 2
        assert r["config"]["carrier_chain"] == "device" and r["config"]["synthesis_kernel"] == "k_synth_ev"
        assert r["value"] > 0 and r["roofline"]["frac"] > 0 and len(r["repeats"]["seconds"]) == 2
        assert len(r["gather"]["per_rank_GBps_to_host"]) == n
        assert r["parity"]["mismatching_blocks"] == 0 and r["parity_checked_blocks"] >= 3 * n
        assert r["parity"]["blocks_digested"] == 2 * 8 * 8
        # every block of every shard cross-checked against the per-sample kernel, by device-side digests
        assert r["parity"]["blocks_cross_checked"] == r["parity"]["blocks_digested"] and r["parity"]["cross_mismatching_blocks"] == 0
        assert r["parity"]["cross_check"]["against"].startswith("k_synth on k_seed")
        assert r["shard_seed_s"] >= r["shard_seed"]["seconds_rank0"] and 0 < r["value_incl_seed"] <= r["value"]
        # the product's node driver over the whole stream on every GPU the process sees (here: one), both layouts, every slot
        # digested inside the sink and compared with what the ranks rendered
        for leg in ("contiguous_indexed", "interleaved_ordered"):
            a = r["node_driver"]["all_gpus"][leg]
            assert a["blocks"] == 2 * 8 * 8 and a["every_block_once"] and a["digests_equal_the_ranks"] and a["blocks_that_differ"] == 0, (n, leg, a)
            assert len(a["shards"]) == len(a["devices"]) >= 1 and all(x["nblocks"] > 0 for x in a["shards"])
        assert r["node_driver"]["all_gpus"]["interleaved_ordered"]["in_stream_order"]
        # the reference's geometry as a time-sharded stream, and the CPU baselines beside the line, at EVERY N
        m = r["m1"]
        assert m["n_gpus"] == n and m["value"] == m["stream"]["value"] > 0 and m["stream"]["synthesis_kernel"] == "k_synth_pd"
        assert m["stream"]["prepass"] == "lap-parallel" and 0 < m["roofline_stream"]["frac"] < 1
        assert m["stream"]["parity"]["mismatching_blocks"] == 0 and m["stream"]["parity"]["checked_blocks"] >= 3 * n
        assert m["stream"]["parity"]["blocks_cross_checked"] == 2 * 8 * 40 and m["stream"]["parity"]["cross_mismatching_blocks"] == 0
        assert r["cpu_baseline"]["value"] > 0 and r["cpu_baseline"]["cores"] == 1 and m["cpu"]["value"] > 0
        # the drop-in call, timed and checked
        for case in ("reference_block_12ch_2.6MSps_300000", "headline_block_16ch_25MSps"):
            f = r["fill_block"][case]
            assert f["equals_oracle"] is True and 0 < f["min_ms"] <= f["median_ms"] <= f["p99_ms"] and f["calls"] == 12
    # the resident legs run at N = 1 only (at N > 1 they would keep N - 1 GPUs idle behind rank 0)
    assert one["m1"]["gpu"]["value"] > 0 and "gpu" not in two["m1"] and "resident" not in two
    assert one["m1"]["stream"]["parity"]["stream_iq_digest"] == two["m1"]["stream"]["parity"]["stream_iq_digest"]
    assert one["m1"]["stream"]["parity"]["stream_end_state_digest"] == two["m1"]["stream"]["parity"]["stream_end_state_digest"]
    assert one["parity"]["cross_check"]["stream_iq_digest"] == two["parity"]["cross_check"]["stream_iq_digest"]
    assert two["shard_seed"]["blocks_before_the_last_shard"] == 64
    assert one["parity"]["stream_end_state_digest"] == two["parity"]["stream_end_state_digest"]
    assert one["config"]["global_samples_per_step"] == two["config"]["global_samples_per_step"]


@pytest.mark.gpu
def test_one_rank_through_the_rccl_path(pkg):
    """What a 1-GPU box can run of the path an 8-GPU node takes: bench.py under torch.distributed.run with ONE rank and the
    nccl (= RCCL) backend, GPSBB_BENCH_FORCE_DIST making it take the N > 1 branches anyway — process group on the device,
    all_reduce of device tensors for the max-over-ranks times, all_gather_object of the digests, barriers — so that the
    driver's multi-GPU run is not the first execution of that code.  The line also carries the node driver's leg."""
    env = dict(os.environ, GPSBB_BENCH_FORCE_DIST="1")
    env.pop("GPSBB_BENCH_BACKEND", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
           "--repeats", "2", "--push-blocks", "8", "--nsamp", "200000", "--depth", "3", "--cpu-budget", "0.3", "--parity-blocks", "2",
           "--parity-spots", "2", "--m1-push-blocks", "40", "--fill-calls", "12"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    r = json.loads(lines[0])
    aff = r["dist"].pop("rank0_affinity")
    assert aff["cpus_bound"] >= 0 and aff["numa_node"] >= -1  # the rank sits on the CPUs local to its GPU where sysfs names them
    assert r["dist"] == {"process_group": True, "backend": "nccl", "world": 1, "hw_queues": 12, "streams_of_the_handle": r["dist"]["streams_of_the_handle"]}
    assert r["n_gpus"] == 1 and r["value"] > 0 and r["parity"]["mismatching_blocks"] == 0
    assert len(r["gather"]["per_rank_GBps_to_host"]) == 1 and r["gather"]["node_GBps_to_host"] > 0
    assert r["node_driver"]["one_shard"]["value"] > 0 and r["node_driver"]["one_shard"]["shards"] == 1
    # what the driver's SCALE run will carry per GPU: the node driver over all visible GPUs, both layouts, validated
    for leg in ("contiguous_indexed", "interleaved_ordered"):
        a = r["node_driver"]["all_gpus"][leg]
        assert a["digests_equal_the_ranks"] and a["every_block_once"] and a["value"] > 0
        assert set(a["shards"][0]) == {"first_block", "nblocks", "device", "numa_node", "cpus_bound", "seed_seconds", "busy_seconds", "wait_seconds"}
    assert r["parity"]["blocks_cross_checked"] == r["parity"]["blocks_digested"] and r["parity"]["cross_mismatching_blocks"] == 0
    # ... and the keys the 2.6 MS/s leg, the CPU baselines and the drop-in call carry at every N
    assert r["m1"]["value"] == r["m1"]["stream"]["value"] > 0 and r["m1"]["stream"]["parity"]["cross_mismatching_blocks"] == 0
    assert r["m1"]["stream"]["parity"]["blocks_cross_checked"] == 2 * 8 * 40 and r["m1"]["roofline_stream"]["kernel"] == "k_synth_pd"
    assert r["cpu_baseline"]["value"] > 0 and r["m1"]["cpu"]["value"] > 0
    assert r["fill_block"]["reference_block_12ch_2.6MSps_300000"]["equals_oracle"] is True


@pytest.mark.gpu
def test_a_wrong_kernel_fails_the_bench(pkg, tmp_path):
    """bench.py's parity check reads blocks of the timed mode's ring back and compares them with the oracle: a build of
    the library whose synthesis kernel flips ONE bit of ONE sample (make broken) makes the run exit non-zero, with the
    mismatch in the line.  The oracle has time for a handful of blocks; EVERY block is cross-checked against the per-sample
    kernel by device-side digests: a second wrong build (make broken2) flips its bit in a block the oracle legs do not visit —
    they pass, the cross-check does not."""
    import subprocess
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "pluto-gps-sim_amd", "csrc"), "broken", "broken2", "VARIANT_DIR=%s" % tmp_path])
    args = ["--steps", "1", "--warmup", "1", "--repeats", "1", "--push-blocks", "8", "--nsamp", "200000", "--depth", "3",
            "--cpu-budget", "0.2", "--parity-blocks", "2", "--m1-push-blocks", "16", "--fill-calls", "4"]
    r = _run(1, args, {"GPSBB_PY_LIB": str(tmp_path / "libgpsbb_broken.so")}, expect_rc=3)
    assert r["parity"]["mismatching_blocks"] >= 1 and r["parity_checked_blocks"] >= 3
    assert r["parity"]["cross_mismatching_blocks"] == 8   # block 1 of each of the 8 pushes
    r = _run(1, args + ["--parity-spots", "2"], {"GPSBB_PY_LIB": str(tmp_path / "libgpsbb_broken2.so")}, expect_rc=3)
    assert r["parity"]["mismatching_blocks"] == 0 and r["parity_checked_blocks"] >= 3   # the oracle legs never look at block 6 of a push
    assert r["parity"]["cross_mismatching_blocks"] == 8 and r["parity"]["blocks_cross_checked"] == 64
