#!/usr/bin/env python3
"""Benchmark of the sparse-search hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

N > 1 without a launcher: this process starts `python -m torch.distributed.run --nproc-per-node N bench.py ...` as a
child BEFORE anything touches the GPU, relays rank 0's JSON line and exits with the child's status (the reference's
launcher line is `deepspeed --num_gpus=4 src/search.py ...`, scripts/search_sparse.sh:14). Under a launcher
(WORLD_SIZE set) it is one rank of N, one GPU per rank.

A "step" is one pass of the hot path over one batch of synthetic queries: the scoring kernels + top-k merge for every
query of the batch, with the index and the CSR queries already resident in HBM.

Headline workload (BASELINE.json metric, configs[1]): Flickr30K-shape text->image sparse search, top-10.
  N > 1: the reference's own data parallelism (src/search.py:180-182): every rank holds the (16 MB) index and scores
  its own queries; no data-path collective; "scaling": "weak".
Extra objects (never part of `value`):
  "c3_coco5k"  configs[2]: COCO-5K shapes, both directions, 1 GPU.
  "c4_1m"      configs[3], the north-star target: 1 M docs / 10 000 queries; at N > 1 the index is doc-range sharded over
               the ranks and the per-shard top-k lists are merged after ONE RCCL all-gather (exact), "scaling": "strong";
               plus the literal term-range partition (every rank resident with its own term range only).
  "c5_hybrid"  configs[4]: dense fp16 MFMA + sparse + min-max fusion, 1 GPU (COCO-5K text->image: 5 000 docs, one tile).
  "c5_hybrid_i2t"  the same search at the shape the reference's own hybrid script runs (scripts/search.sh: 25 010 caption
               docs, 5 000 image queries, --remove_query): four tiles.

Failure policy: a rank that cannot get its own GPU, an RCCL communicator that does not come up (unless
--allow-gloo-fallback), or an exchange that does not finish within --c4-timeout end the run with a NON-ZERO status;
the JSON line is still printed, with "error" / "hung" fields saying what happened.

rank 0 prints ONE JSON line. torch is launcher plumbing only (gloo barrier / max-reduce, the contract's
torch.cuda.synchronize); the search path is ctypes -> libmsr.so -> HIP. libmsr.so is loaded BEFORE torch so that the
process resolves ONE HIP runtime and ONE librccl — the /opt/rocm ones libmsr.so was compiled against (torch ships
older copies under torch/lib with the same sonames; whichever is loaded first serves both).
"""
from __future__ import annotations

import argparse
import json
import os
import re
import socket
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# ceilings from MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0    # HBM3E spec peak; ~6.3 TB/s is what a copy achieves
L2_PEAK_GBS = 34500.0    # aggregate L2 -> CU bandwidth (8 XCDs)
MFMA_F16_PEAK_TF = 2500.0  # dense fp16/bf16 MFMA peak
N_SIMDS = 1024           # 256 CUs x 4 SIMDs
N_XCDS = 8
COUNTERS_FILE = os.path.join(ROOT, "profiles", "r03_counters.json")
LDS_PEAK_GBS = 256 * 128 * 2.4  # 128 B / clk / CU x 256 CUs x 2.4 GHz = 78.6 TB/s (guide figure; the measured one is used)

EXIT_LAUNCH = 2   # bad launch: fewer GPUs than ranks, WORLD_SIZE / --gpus mismatch
EXIT_HUNG = 3     # the multi-rank exchange did not finish (watchdog)
EXIT_COMM = 4     # RCCL communicator did not come up / exchange failed

_PHASE = "start"


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def phase(name, rank=None):
    """Per-rank progress marker on stderr: when a run hangs, the last marker of every rank says where."""
    global _PHASE
    _PHASE = name
    log(f"[bench r{os.environ.get('RANK', '0') if rank is None else rank}] phase: {name}")


# ------------------------------------------------------------------------------------------------ launcher
def parse_args(argv):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--flickr-images", type=int, default=31014)
    ap.add_argument("--tile-docs", type=int, default=0)
    ap.add_argument("--c4-docs", type=int, default=1_000_000)
    ap.add_argument("--c4-queries", type=int, default=10_000)
    ap.add_argument("--c4-tile-docs", type=int, default=0)
    ap.add_argument("--dense-max", type=int, default=-1, help="index build option dense_max_terms (-1: library default)")
    ap.add_argument("--dense-density", type=float, default=-1.0, help="index build option dense_min_density")
    ap.add_argument("--c4-timeout", type=float, default=900.0, help="watchdog for the 1 M-doc object at N > 1 (s)")
    ap.add_argument("--no-c3", action="store_true", help="skip the COCO-5K (config 3) extra object")
    ap.add_argument("--only-c3", action="store_true", help="(profiling) run only the COCO-5K workloads")
    ap.add_argument("--c3-dir", choices=["both", "i2t", "t2i"], default="both", help="(profiling) one direction only")
    ap.add_argument("--no-c4", action="store_true", help="skip the 1 M-doc extra object")
    ap.add_argument("--no-term-shards", action="store_true", help="skip the term-range sharded variant at N > 1")
    ap.add_argument("--only-c4", action="store_true", help="(profiling) run only the 1 M-doc workload")
    ap.add_argument("--c5-docs", type=int, default=5000)
    ap.add_argument("--c5-queries", type=int, default=25010)
    ap.add_argument("--no-c5", action="store_true", help="skip the hybrid (config 5) extra objects")
    ap.add_argument("--c5-shape", choices=["both", "t2i", "i2t"], default="both",
                    help="(profiling) one hybrid shape only: t2i = configs[4], i2t = the reference's own script")
    ap.add_argument("--only-c5", action="store_true", help="(profiling) run only the hybrid workload")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline / parity samples")
    ap.add_argument("--cpu-seconds", type=float, default=8.0, help="CPU work per baseline row (s)")
    ap.add_argument("--cpu-threads", type=int, default=min(16, os.cpu_count() or 1))
    ap.add_argument("--host-threads", type=int, default=min(16, os.cpu_count() or 1))
    ap.add_argument("--allow-gloo-fallback", action="store_true",
                    help="if RCCL cannot be initialised, move the per-shard lists through gloo instead of failing")
    ap.add_argument("--allow-oversubscribe", action="store_true",
                    help="(rehearsal on a small box) let several ranks share one GPU instead of failing")
    ap.add_argument("--launcher-selftest", action="store_true",
                    help="ranks only rendezvous (gloo), barrier, max-reduce and print; no GPU work (CPU test of the launcher)")
    return ap.parse_args(argv)


def visible_gpus():
    import torch  # device_count() does not initialise the GPU on this image

    return int(torch.cuda.device_count())


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_ranks(args, argv):
    """--gpus N > 1 outside a launcher: start N ranks as a child process group and relay rank 0's line."""
    if not args.launcher_selftest and not args.allow_oversubscribe:
        n_dev = visible_gpus()
        if n_dev < args.gpus:
            log(f"[bench] ERROR: --gpus {args.gpus} but only {n_dev} HIP device(s) are visible; refusing to stack ranks on "
                f"one GPU (a 1-GPU number must not be reported as n_gpus={args.gpus}). "
                f"Use --allow-oversubscribe only for a plumbing rehearsal.")
            return EXIT_LAUNCH
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + list(argv)
    log("[bench] launching: " + " ".join(cmd))
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    p = subprocess.run(cmd, stdout=subprocess.PIPE, env=env)
    lines = [ln for ln in p.stdout.decode("utf-8", "replace").splitlines() if ln.strip().startswith("{")]
    rc = p.returncode
    if lines:
        print(lines[-1], flush=True)
        try:  # torch.distributed.run folds every rank failure into 1: hand on the ranks' own status
            rc = int(json.loads(lines[-1]).get("exit_status", rc)) if rc else rc
        except Exception:
            pass
    elif rc == 0:
        log("[bench] ERROR: the ranks exited 0 but printed no JSON line")
        return 1
    return rc


class Ranks:
    """Launcher plumbing: rank info, barrier and max-over-ranks (gloo; the data path never touches torch)."""

    def __init__(self, args):
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.dist = None
        if args.gpus != self.world:
            log(f"[bench r{self.rank}] ERROR: --gpus {args.gpus} but WORLD_SIZE={self.world}: launch with "
                f"--nproc-per-node {args.gpus} (or run `python bench.py --gpus {args.gpus}` and let it launch the ranks)")
            sys.exit(EXIT_LAUNCH)
        if not args.launcher_selftest:
            n_dev = visible_gpus()
            if self.local_rank >= n_dev:
                if args.allow_oversubscribe and n_dev > 0:
                    log(f"[bench r{self.rank}] REHEARSAL: local rank {self.local_rank} shares GPU {self.local_rank % n_dev}")
                    self.local_rank %= n_dev
                else:
                    log(f"[bench r{self.rank}] ERROR: local rank {self.local_rank} has no GPU of its own "
                        f"({n_dev} visible)")
                    sys.exit(EXIT_LAUNCH)
        if self.world > 1:
            import datetime

            import torch.distributed as dist

            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group("gloo", timeout=datetime.timedelta(minutes=20))
            self.dist = dist

    def barrier(self):
        if self.dist:
            self.dist.barrier()

    def max(self, x):
        if not self.dist:
            return x
        import torch

        t = torch.tensor([x], dtype=torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def gather_objects(self, obj):
        if not self.dist:
            return [obj]
        out = [None] * self.world
        self.dist.all_gather_object(out, obj)
        return out

    def bcast_bytes(self, b, n):
        if not self.dist:
            return b
        obj = [b if self.rank == 0 else None]
        self.dist.broadcast_object_list(obj, src=0)
        return obj[0]

    def any_failed(self, failed):
        """True on every rank if any rank reports a failure (keeps the ranks in step instead of deadlocking)."""
        return self.max(1.0 if failed else 0.0) > 0

    def close(self):
        if self.dist:
            self.dist.destroy_process_group()


_SYNC_DEVICE = 0
_M = None


def device_sync():
    """Contract: a device-wide fence on both sides of the timed region, on THIS rank's GPU: hipDeviceSynchronize
    through libmsr.so (covers libmsr's own stream) and torch.cuda.synchronize() as the contract names it. No other
    torch GPU call is made in this process (tests/test_cabi.py greps for it): torch's bundled kernels expect torch's
    own (older) HIP runtime, while the process runs on the /opt/rocm runtime libmsr.so was built for — round 2's 20
    segfaults under rocprofv3 were torch device-to-device copies on that foreign runtime."""
    _M.device_sync(_SYNC_DEVICE)
    try:
        import torch

        if torch.cuda.device_count() > _SYNC_DEVICE:   # (device_count does not initialise the GPU; is_available does)
            torch.cuda.synchronize(_SYNC_DEVICE)
    except Exception:
        pass


_HBM_COPY_GBS = None


def hbm_copy_gbs():
    """Measured HBM bandwidth of a 1 GiB device-to-device copy on this rank's GPU (read + write bytes / time), quoted
    beside the vendor peak as SURVEY.md §8d asks (msr_device_copy_gbs: hipMemcpyAsync + HIP events, no torch)."""
    global _HBM_COPY_GBS
    if _HBM_COPY_GBS is None:
        try:
            _HBM_COPY_GBS = round(_M.device_copy_gbs(_SYNC_DEVICE), 1)
        except Exception as e:
            log(f"[bench] device copy measurement failed: {e}")
            _HBM_COPY_GBS = 0.0
    return _HBM_COPY_GBS or None


def timed_steps(batch, k, steps, warmup, ranks, sharded=False, step=None):
    """`step` (optional) replaces batch.search for exchanges that run on the host (gloo fallback)."""
    run = step if step is not None else (lambda: batch.search(k, sharded=sharded))
    for _ in range(warmup):
        run()
    batch.sync()
    batch.timing_reset()
    ranks.barrier()
    device_sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        run()
    batch.sync()
    device_sync()
    ranks.barrier()
    dt = time.perf_counter() - t0
    calls, score_ms, merge_ms = batch.timing_sum()
    if os.environ.get("MSR_DEBUG_FLAGS"):
        st = batch.debug_stamps().astype(np.float64)
        if st.sum() > 0:
            names = ["init", "stage", "stream", "wait", "maxima", "cand", "rank", "resolve"]
            log("[bench] wave-0 phase shares: " + ", ".join(f"{n}={v / st.sum():.3f}" for n, v in zip(names, st))
                + f"; cycles/WG-launch total={st.sum() / max(calls + warmup, 1):.3e}")
    return ranks.max(dt), score_ms / max(calls, 1), merge_ms / max(calls, 1)


# ------------------------------------------------------------------------------------------------ roofline
_COUNTERS = None
_COUNTERS_STATE = None


def counters_state():
    """Is profiles/r03_counters.json about THIS binary? The file carries the kernel-source and code-object hashes of the
    build it was collected on (scripts/prof_counters.py); when they differ from this tree's, every counter-backed figure
    is withheld and the bench line says `counters_stale`."""
    global _COUNTERS, _COUNTERS_STATE
    if _COUNTERS_STATE is None:
        from mllm_sparse_retrieval_amd import _buildinfo

        try:
            _COUNTERS = json.load(open(COUNTERS_FILE))
        except Exception as e:
            _COUNTERS = {}
            _COUNTERS_STATE = {"file": os.path.relpath(COUNTERS_FILE, ROOT), "stale": True, "reason": f"unreadable: {e}"}
            return _COUNTERS_STATE
        st = _COUNTERS.get("_stamp")
        reason = _buildinfo.stale_reason(st)
        _COUNTERS_STATE = {"file": os.path.relpath(COUNTERS_FILE, ROOT), "stale": reason is not None,
                           "collected_at_git_head": (st or {}).get("git_head"),
                           "kernel_source_sha256": (st or {}).get("kernel_source_sha256")}
        if reason:
            _COUNTERS_STATE["reason"] = reason
            _COUNTERS = {}
    return _COUNTERS_STATE


def counters(workload, kernel_re):
    """Per-STEP counter sums of the kernels matching `kernel_re` in one profiled workload, from the committed
    rocprofv3 --pmc passes (profiles/r03_counters.json, written by scripts/prof_counters.py; one pass per counter
    group, as MI355X_MICROARCH.md prescribes). None when the workload was not profiled or the file is stale."""
    counters_state()
    w = _COUNTERS.get(workload)
    if not w:
        return None
    tot = {}
    for name, c in w.get("kernels", {}).items():
        if re.search(kernel_re, name):
            for key, v in c.items():
                tot[key] = tot.get(key, 0.0) + float(v)
    if not tot:
        return None
    steps = float(w.get("steps", 1))
    return {key: v / steps for key, v in tot.items()}


_PEAKS = None


def pipe_peaks():
    """The pipes the scorer's useful work runs on, MEASURED on this rank's GPU at the kernel's launch shape
    (msr_device_peak_rates: kernels that issue nothing but ds_add_u32 / v_dot2_u32_u16 / the accumulator tile's
    ds_write_b128 + 2 x ds_read_b128)."""
    global _PEAKS
    if _PEAKS is None:
        try:
            _PEAKS = _M.device_peak_rates(_SYNC_DEVICE)
        except Exception as e:
            log(f"[bench] peak-rate measurement failed: {e}")
            _PEAKS = {}
    return _PEAKS


def binding_fractions(c, kernel_ms):
    """The ceilings a kernel can actually run into, as fractions <= 1, from per-step counters `c` and the LIVE kernel
    time per step: HBM-side traffic vs 8 TB/s, L2 hits x 128 B vs 34.5 TB/s, VALU issue (quad-cycle busy count vs the
    SIMD-cycles of the profiled dispatches), plus scalar instructions per vector instruction."""
    out = {}
    t = kernel_ms * 1e-3
    if c is None or t <= 0:
        return out
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        # KiB units; FETCH_SIZE counts 64 B per 128-B request on gfx950 -> x2 (MI355X_MICROARCH.md, HBM section)
        traffic = (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0
        out["traffic"] = int(traffic)
        out["hbm_frac"] = round(traffic / t / (HBM_PEAK_GBS * 1e9), 4)
    if "TCC_HIT_sum" in c:
        out["l2_hit_bytes"] = int(c["TCC_HIT_sum"] * 128)
        out["l2_frac"] = round(c["TCC_HIT_sum"] * 128.0 / t / (L2_PEAK_GBS * 1e9), 4)
        if "TCC_MISS_sum" in c:
            out["l2_hit_rate"] = round(c["TCC_HIT_sum"] / max(c["TCC_HIT_sum"] + c["TCC_MISS_sum"], 1.0), 4)
    if "SQ_ACTIVE_INST_VALU" in c and "GRBM_GUI_ACTIVE" in c:
        # busy quad-cycles of the VALU (a property of the work done) against the SIMD-cycles of the LIVE kernel time at
        # the clock the counter pass saw (GRBM_GUI_ACTIVE is summed over the 8 XCDs; a counter set can slow a kernel
        # down, so the clock comes from that pass's own kernel duration)
        cycles_pass = c["GRBM_GUI_ACTIVE"] / N_XCDS
        dur_pass = c.get("duration_ns_sq2")
        if dur_pass:
            # (GRBM_GUI_ACTIVE over-counts for a dispatch that follows another kernel closely — hybrid_tiles after the
            # GEMM reads 7 "GHz" — so the clock is capped at the 2.4 GHz peak engine clock: busy fractions can only
            # come out LOWER than the truth that way, never higher)
            clock_ghz = min(cycles_pass / dur_pass, 2.4)
            cycles = t * clock_ghz * 1e9
            out["clock_ghz"] = round(clock_ghz, 3)
        else:
            cycles = cycles_pass
        out["valu_busy"] = round(4.0 * c["SQ_ACTIVE_INST_VALU"] / max(cycles * N_SIMDS, 1.0), 4)
        if "SQ_ACTIVE_INST_LDS" in c:
            out["lds_issue_busy"] = round(4.0 * c["SQ_ACTIVE_INST_LDS"] / max(cycles * N_SIMDS, 1.0), 4)
    if "SQ_INSTS_SALU" in c and "SQ_INSTS_VALU" in c:
        out["salu_per_valu"] = round(c["SQ_INSTS_SALU"] / max(c["SQ_INSTS_VALU"], 1.0), 3)
    return out


def sparse_roofline(batch, k, score_ms_avg, workload_name, kernel_re="score_tiles"):
    """roofline object of the sparse scorer.

    frac = USEFUL-WORK roofline: the least time the step's work needs on the pipe that binds it, over the kernel time
    measured live. The work is what any (tile, query) decomposition of this algorithm has to do, each part priced at the
    MEASURED peak of the instruction that does it (pipe_peaks):
        lds_atomic  inverted-list postings        / ds_add_u32 lane-ops per second
        valu_dot2   dense-head postings / 2       / v_dot2_u32_u16 lane-ops per second
        lds_bw      accumulator tiles: init write + the selection's two reads  / LDS bytes per second at that mix
        hbm         HBM-side bytes (counters)     / 8 TB/s
    `utilisation` keeps the pipe-busy figures of round 2 (how busy, not how useful), `algorithmic` SURVEY §8d's number."""
    by, postings = batch.algo_bytes(k)
    t = score_ms_avg * 1e-3
    algo_gbs = by / t / 1e9 if t > 0 else 0.0
    state = counters_state()
    fr = binding_fractions(counters(workload_name, kernel_re), score_ms_avg)
    work = batch.work()
    pk = pipe_peaks()
    r = {"kernel": "score_tiles", "kernel_ms": round(score_ms_avg, 4),
         "launches_per_step": 2 if batch.index.n_tiles >= 2 else 1}
    min_ms = {}
    if pk.get("ds_add_per_s"):
        min_ms["lds_atomic"] = work["sparse_postings"] / pk["ds_add_per_s"] * 1e3
        min_ms["valu_dot2"] = work["dense_head_postings"] / 2.0 / pk["dot2_per_s"] * 1e3
        min_ms["lds_bw"] = (work["acc_init_bytes"] + work["acc_select_bytes"]) / pk["lds_bytes_per_s"] * 1e3
    if fr.get("traffic") is not None:
        min_ms["hbm"] = fr["traffic"] / (HBM_PEAK_GBS * 1e9) * 1e3
    complete = len(min_ms) == 4 and not state["stale"] and t > 0
    if min_ms:
        bound = max(min_ms, key=min_ms.get)
        useful = min_ms[bound] / score_ms_avg if score_ms_avg > 0 else None
        achieved, peak, unit = {
            "lds_atomic": (work["sparse_postings"] / t / 1e9, pk.get("ds_add_per_s", 0) / 1e9, "G postings/s (ds_add_u32 lane-ops)"),
            "valu_dot2": (work["dense_head_postings"] / 2.0 / t / 1e9, pk.get("dot2_per_s", 0) / 1e9, "G v_dot2_u32_u16 lane-ops/s"),
            "lds_bw": ((work["acc_init_bytes"] + work["acc_select_bytes"]) / t / 1e9, pk.get("lds_bytes_per_s", 0) / 1e9, "GB/s (LDS)"),
            "hbm": ((fr.get("traffic") or 0) / t / 1e9, HBM_PEAK_GBS, "GB/s"),
        }[bound]
        r.update(bound=bound, achieved=round(achieved, 1), peak=round(peak, 1), unit=unit,
                 frac=round(useful, 4) if complete else None)
        r["useful"] = {"min_ms_per_pipe": {b: round(v, 4) for b, v in min_ms.items()},
                       "frac_if_counters_were_current": None if complete else (round(useful, 4) if useful else None),
                       "work_per_step": work,
                       "peaks_measured": {"ds_add_u32_lane_ops_per_s": pk.get("ds_add_per_s"),
                                          "v_dot2_u32_u16_lane_ops_per_s": pk.get("dot2_per_s"),
                                          "lds_bytes_per_s_1w2r": pk.get("lds_bytes_per_s"), "cus": pk.get("cus"),
                                          "how": "msr_device_peak_rates on this GPU, in this run (profiles/r03_peak_rates.txt)"},
                       "note": "frac = the binding pipe's minimum time / kernel time: what share of the kernel's life the "
                               "postings' own adds / dot products / accumulator traffic would need at the measured "
                               "instruction peaks; the rest is per-(tile, query) fixed cost, address arithmetic and waits"}
        if complete is False and r["useful"]["frac_if_counters_were_current"] is None:
            r["useful"].pop("frac_if_counters_were_current")
    else:
        r.update(bound="hbm", frac=None, achieved=None, peak=HBM_PEAK_GBS, unit="GB/s")
    r["counters_stale"] = bool(state["stale"])
    r["traffic"] = fr.get("traffic")
    r["utilisation"] = {key: fr.get(key) for key in ("hbm_frac", "l2_frac", "valu_busy", "salu_per_valu", "lds_issue_busy",
                                                      "l2_hit_rate", "clock_ghz")}
    r["algorithmic"] = {"bytes_per_step": by, "postings_per_step": postings, "gbps": round(algo_gbs, 1),
                        "over_hbm_peak": round(algo_gbs / HBM_PEAK_GBS, 4),
                        "note": "SURVEY §8d bytes (6 B per posting, every query streams privately) / kernel time: NOT a "
                                "ceiling for a query-batched scorer — a tile's postings are shared by the queries in "
                                "flight through L2, stored as 4-byte postings / 2-byte dense-head weights"}
    r["hbm_copy_measured"] = hbm_copy_gbs()
    r["counters"] = state
    return r


# ------------------------------------------------------------------------------------------------ CPU baseline
def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(wl, got, target_seconds, threads, rows=True):
    """The oracle's C restatement (multithreaded term-at-a-time) timed on this host on a bounded query sample, and
    used as the checker of the GPU results on that sample. Checker only: nothing here feeds the GPU path.
    Rows: 16 threads on one big batch (the reference's --threads 16, scripts/search_sparse.sh:17) is
    the headline `value`; beside it every CPU this job may use, and the reference's own call shape of 4 queries per
    batch_search call (scripts/search_sparse.sh:16)."""
    from oracle import taat

    dp, dt, dw = wl.docs
    qp, qt, qw = wl.queries
    nq = len(qp) - 1
    t0 = time.perf_counter()
    oix, _ = taat.TaatIndex.from_rows_by_docid(dp, dt, dw, wl.n_terms)
    build_s = time.perf_counter() - t0

    def sub(a, b):
        return (qp[a: b + 1] - qp[a]), qt[qp[a]: qp[b]], qw[qp[a]: qp[b]]

    def sized(thr, per_call=None, mode="exhaustive"):
        """queries that fill ~target_seconds at this setting (from a short probe)"""
        nonlocal target_seconds
        probe = min(nq, max(4 * thr, 64))
        t1 = time.perf_counter()
        if per_call:
            for a in range(0, probe, per_call):
                oix.search(*sub(a, min(a + per_call, probe)), wl.k, threads=thr)
        else:
            oix.search(*sub(0, probe), wl.k, threads=thr, mode=mode)
        per_q = (time.perf_counter() - t1) / probe
        return int(min(nq, max(probe, target_seconds / max(per_q, 1e-9))))

    n = sized(threads)
    t0 = time.perf_counter()
    w_ord, w_sc, w_n = oix.search(*sub(0, n), wl.k, threads=threads)
    dt_s = time.perf_counter() - t0
    g_ord, _, g_u32, g_n = got
    mask = np.arange(wl.k)[None, :] < w_n[:, None]
    mism = int((g_n[:n] != w_n).sum() + (g_ord[:n].astype(np.int64)[mask] != w_ord[mask]).sum()
               + (g_u32[:n].astype(np.int64)[mask] != w_sc[mask]).sum())
    affinity = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cb = {"value": round(n / dt_s, 1), "unit": "queries/s", "cores": threads, "kind": "port",
          "cpu_model": cpu_model(), "host_logical_cpus": os.cpu_count(), "affinity_cpus": affinity,
          "sample": f"first {n} of {nq} queries of the same workload, {threads} threads, one call, exhaustive "
                    f"term-at-a-time C restatement (oracle/oracle_taat.c); index build {build_s:.1f}s not timed",
          "seconds": round(dt_s, 2)}
    if rows:
        extra = []
        thr_all = max(1, min(affinity, 256))
        if thr_all != threads:
            # every CPU this job may run on. The accumulators are scanned and cleared over the docs a query touched only
            # (mode "touched": with a full 4-byte-per-doc scan + memset per query, 256 workers thrash the caches and run
            # SLOWER than 16 — round 2's row); same hits as the exhaustive scan (tests/test_oracle.py)
            n2 = sized(thr_all, mode="touched")
            t0 = time.perf_counter()
            r2 = oix.search(*sub(0, n2), wl.k, threads=thr_all, mode="touched")
            d2 = time.perf_counter() - t0
            nn = min(n, n2)
            extra.append({"value": round(n2 / d2, 1), "cores": thr_all, "queries": n2, "seconds": round(d2, 2),
                          "kind": "port", "identical_to_exhaustive": bool((r2[0][:nn] == w_ord[:nn]).all() and (r2[1][:nn] == w_sc[:nn]).all()),
                          "shape": f"one call, {thr_all} threads = every CPU this job may run on (sched_getaffinity); "
                                   f"term-at-a-time, only touched docs scanned and cleared"})
        # thread-count scan of the exhaustive port (the strongest row is the honest "what this host can do" figure;
        # `value` stays at the reference's --threads 16)
        scan = []
        for thr in (32, 64, 128):
            if thr < thr_all and thr != threads:
                save, target_seconds = target_seconds, min(target_seconds, 2.0)
                n5 = sized(thr)
                target_seconds = save
                t0 = time.perf_counter()
                oix.search(*sub(0, n5), wl.k, threads=thr)
                d5 = time.perf_counter() - t0
                scan.append({"cores": thr, "value": round(n5 / max(d5, 1e-9), 1), "queries": n5, "seconds": round(d5, 2)})
        if scan:
            best = max(scan + [{"cores": threads, "value": cb["value"]}], key=lambda r: r["value"])
            extra.append({"kind": "port", "shape": "thread-count scan of the exhaustive port, one call each", "scan": scan,
                          "best": {"cores": best["cores"], "value": best["value"]}})
        # a PRUNING engine on the same host: document-at-a-time MaxScore in 4096-doc windows (the strategy of Lucene's
        # bulk scorer for pure disjunctions), exact, same file. With ~flat learned-sparse weights few lists ever become
        # non-essential at k = 10, so pruning does not beat the exhaustive scan here: reported, not assumed.
        n4 = sized(threads, mode="maxscore")
        t0 = time.perf_counter()
        r4 = oix.search(*sub(0, n4), wl.k, threads=threads, mode="maxscore")
        d4 = time.perf_counter() - t0
        nn = min(n, n4)
        extra.append({"value": round(n4 / max(d4, 1e-9), 1), "cores": threads, "queries": n4, "seconds": round(d4, 2),
                      "kind": "port+pruning",
                      "identical_to_exhaustive": bool((r4[0][:nn] == w_ord[:nn]).all() and (r4[1][:nn] == w_sc[:nn]).all()),
                      "shape": f"one call, {threads} threads, document-at-a-time MaxScore (oracle_taat.c mode 2): exact top-{wl.k}"})
        n3 = sized(threads, per_call=4) // 4 * 4
        t0 = time.perf_counter()
        for a in range(0, n3, 4):
            oix.search(*sub(a, a + 4), wl.k, threads=threads)
        d3 = time.perf_counter() - t0
        extra.append({"value": round(n3 / max(d3, 1e-9), 1), "cores": threads, "queries": n3, "seconds": round(d3, 2),
                      "shape": f"4 queries per call, {threads} threads (the reference's per_device_batch_size 4 / "
                               f"--threads 16, scripts/search_sparse.sh:16-17)"})
        cb["rows"] = extra
        # the strongest CPU figure of this host over every row (one big call): what the GPU number should be held against
        one_call = [cb["value"]] + [r["value"] for r in extra if "value" in r and "4 queries per call" not in r.get("shape", "")]
        one_call += [r["best"]["value"] for r in extra if "best" in r]
        cb["best_one_call_value"] = max(one_call)
    return cb, {"checked_queries": n, "mismatches": mism}


# ------------------------------------------------------------------------------------------------ workloads
def run_headline(args, ranks, m, wlmod):
    seed = 1
    t0 = time.perf_counter()
    wl = wlmod.flickr30k_t2i(n_images=args.flickr_images, seed=seed, threads=args.host_threads)
    if ranks.world > 1:  # every rank scores its own captions (DP over queries, src/search.py:180-182)
        wl.queries = wlmod.planted_captions(wl.docs, wl.n_terms, 5, 8, 15, 6, seed + 1 + 100 * ranks.rank,
                                            args.host_threads)
    log(f"[bench r{ranks.rank}] workload {wl.name} generated in {time.perf_counter() - t0:.1f}s")
    tmp = tempfile.mkdtemp(prefix="msr_bench_")
    path = m.build_index_from_csr(os.path.join(tmp, f"flickr_{ranks.rank}.idx"), *wl.docs, wl.n_terms,
                                  threads=args.host_threads, tile_docs=args.tile_docs)
    ix = m.SparseIndex(path, device=ranks.local_rank)
    qp, qt, qw = wl.queries
    nq = len(qp) - 1
    batch = ix.batch(qp, qt, qw, wl.k)
    phase("headline timed region")
    wall, score_ms, merge_ms = timed_steps(batch, wl.k, args.steps, args.warmup, ranks)
    out = {
        "metric": "queries/sec, Flickr30K-shape text->image sparse search (top-10), MI355X",
        "value": round(ranks.world * nq * args.steps / wall, 1),
        "unit": "queries/s",
        "n_gpus": ranks.world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(wall / args.steps * 1e3, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u32",
        "data": "synthetic",
        "config": {"workload": wl.description, "queries_per_step_per_gpu": nq, "k": wl.k,
                   "tile_docs": ix.tile_docs, "n_tiles": ix.n_tiles,
                   "parallelism": "1 GPU" if ranks.world == 1 else f"dp{ranks.world} over queries, index replicated"},
        "parity_status": "partial: bit-exact against this repo's CPU restatement (oracle/), which restates a DECLARED "
                         "contract of pyserini/Lucene impact search; the reference holds no fixtures and Lucene cannot "
                         "run here, so the scorer's parity is unpinned (DESIGN.md §3)",
        "runtime": m.runtime_info(),
    }
    devices = ranks.gather_objects({"rank": ranks.rank, "device": ix.device})
    if ranks.rank == 0:
        out["rank_devices"] = devices
        out["roofline"] = sparse_roofline(batch, wl.k, score_ms, wl.name)
        out["roofline"]["merge_kernel_ms"] = round(merge_ms, 4)
        got = batch.fetch()
        # Recall@1/5/10 of the GPU results (qrels: caption j <-> image j // 5)
        doc_int = np.array([int(ix.docid(o)) for o in range(ix.n_docs)], dtype=np.int64)
        ranked = np.where(np.arange(wl.k)[None, :] < got[3][:, None], doc_int[np.minimum(got[0], ix.n_docs - 1)], -1)
        target = (np.arange(nq) // 5)[:, None]
        out["recall"] = {f"R@{k}": round(float((ranked[:, :k] == target).any(axis=1).mean()), 5) for k in (1, 5, 10)}
        if ranks.world == 1 and not args.no_cpu:
            phase("headline cpu baseline")
            cb, par = cpu_baseline(wl, got, args.cpu_seconds, args.cpu_threads)
            out["cpu_baseline"] = cb
            out["parity"] = par
            out["speedup_vs_cpu"] = round(out["value"] / cb["value"], 1)
            out["speedup_vs_best_cpu_row"] = round(out["value"] / cb["best_one_call_value"], 1)
        if ranks.world == 1 and not args.no_cpu:  # (--no-cpu = profiling runs: only the timed launches)
            # PCIe-inclusive figures (never `value`): host CSR in -> host results out through msr_search_csr, and the
            # reference's own call shape of 4 queries per batch_search call (scripts/search_sparse.sh:16)
            ix.search_csr(qp, qt, qw, wl.k)
            t0 = time.perf_counter()
            ix.search_csr(qp, qt, qw, wl.k)
            e2e = time.perf_counter() - t0
            q4 = (qp[:5] - qp[0]), qt[: qp[4]], qw[: qp[4]]
            for _ in range(20):
                ix.search_csr(*q4, wl.k)
            t0 = time.perf_counter()
            for _ in range(200):
                ix.search_csr(*q4, wl.k)
            small = (time.perf_counter() - t0) / 200
            laps = None   # where the microseconds of such a call go (msr_search_laps; averaged over 200 more calls)
            for _ in range(200):
                ix.search_csr(*q4, wl.k)
                one = m.search_laps()
                laps = one if laps is None else {key: laps[key] + v for key, v in one.items()}
            laps = {key: round(v / 200, 2) for key, v in laps.items() if not key.endswith("_kernel")}
            laps["python_and_ctypes"] = round(small * 1e6 - laps["call_total"], 2)
            # ... and the drop-in class itself on query STRINGS (tokens repeated weight times, src/search.py:419-422)
            from mllm_sparse_retrieval_amd.compat import LuceneImpactSearcher

            strings = [" ".join(" ".join([str(int(t))] * int(w)) for t, w in zip(qt[qp[i]:qp[i + 1]], qw[qp[i]:qp[i + 1]]))
                       for i in range(4)]
            searcher = LuceneImpactSearcher(path, None, device=ranks.local_rank)
            searcher.batch_search(strings, ["0", "1", "2", "3"], wl.k, threads=16)
            t0 = time.perf_counter()
            for _ in range(50):
                searcher.batch_search(strings, ["0", "1", "2", "3"], wl.k, threads=16)
            dropin = (time.perf_counter() - t0) / 50
            searcher.close()
            out["host_inclusive"] = {"queries_per_s_one_call": round(nq / e2e, 1),
                                     "ms_per_call_4_queries": round(small * 1e3, 4),
                                     "us_breakdown_4_queries": laps,
                                     "us_breakdown_note": "prepare_upload = query normalisation on the host + writing the CSR "
                                     "into mapped pinned memory (the kernels read it over PCIe: no upload on the stream); "
                                     "enqueue_kernels = two launches (score_tiles, merge_small); wait_stream = launch latency "
                                     "+ both kernels + completion (two dependent EMPTY kernels + sync measure 16.0 us on this "
                                     "box, profiles/r03_latency_lab.txt); download = copy out of the mapped result block",
                                     "queries_per_s_4_per_call": round(4 / small, 1),
                                     "ms_per_batch_search_4_query_strings": round(dropin * 1e3, 4),
                                     "tokens_per_query_string": round(sum(len(x.split()) for x in strings) / 4, 1)}
    batch.close()
    ix.close()
    try:
        os.remove(path)
        os.rmdir(tmp)
    except OSError:
        pass
    return out


def run_c3(args, ranks, m, wlmod):
    """configs[2]: COCO-5K both directions, ~120-nnz queries, V = 30 000, top-10, one GPU."""
    out = {}
    for direction in (("i2t", "t2i") if args.c3_dir == "both" else (args.c3_dir,)):
        wl = wlmod.coco5k(direction, threads=args.host_threads)
        tmp = tempfile.mkdtemp(prefix="msr_c3_")
        path = m.build_index_from_csr(os.path.join(tmp, "c3.idx"), *wl.docs, wl.n_terms, threads=args.host_threads)
        ix = m.SparseIndex(path, device=ranks.local_rank)
        qp, qt, qw = wl.queries
        nq = len(qp) - 1
        batch = ix.batch(qp, qt, qw, wl.k)
        phase(f"c3 {direction} timed region")
        wall, score_ms, merge_ms = timed_steps(batch, wl.k, args.steps, args.warmup, ranks)
        o = {"workload": wl.description + ", top-10", "value": round(nq * args.steps / wall, 1), "unit": "queries/s",
             "ms_per_step": round(wall / args.steps * 1e3, 4), "n_tiles": ix.n_tiles, "tile_docs": ix.tile_docs,
             "roofline": sparse_roofline(batch, wl.k, score_ms, wl.name)}
        o["roofline"]["merge_kernel_ms"] = round(merge_ms, 4)
        if not args.no_cpu:
            cb, par = cpu_baseline(wl, batch.fetch(), min(args.cpu_seconds, 4.0), args.cpu_threads, rows=False)
            o["cpu_baseline"] = cb
            o["parity"] = par
        out[direction] = o
        batch.close()
        ix.close()
        try:
            os.remove(path)
            os.rmdir(tmp)
        except OSError:
            pass
    return out


def run_c4(args, ranks, m, wlmod, status):
    """configs[3]: 1 M docs; doc-range shards + one RCCL all-gather of the per-shard top-k (exact); then the literal
    term-range partition (every rank resident with its own term range only; exchange = reduce-scatter of accumulators)."""
    t0 = time.perf_counter()
    shm = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else tempfile.gettempdir()
    path = os.path.join(shm, f"msr_c4_{os.environ.get('MASTER_PORT', 'p')}_{os.getuid()}.idx")
    wl = None
    err = None
    if ranks.rank == 0:
        try:
            phase("c4 corpus + index build")
            wl = wlmod.c4_1m(n_docs=args.c4_docs, n_queries=args.c4_queries, threads=args.host_threads)
            m.build_index_from_csr(path, *wl.docs, wl.n_terms, threads=args.host_threads, tile_docs=args.c4_tile_docs)
            log(f"[bench] c4 corpus + index in {time.perf_counter() - t0:.1f}s -> {path}")
        except Exception as e:
            err = e
    if ranks.any_failed(err is not None):
        raise RuntimeError(f"c4 corpus/index build failed on rank 0: {err}")
    try:
        ix = batch = None
        sharded = ranks.world > 1
        try:
            qp, qt, qw = m.synth_vectors(args.c4_queries, 120, 30000, seed=4, threads=args.host_threads)
            qp, qt, qw = qp.astype(np.int64), qt.astype(np.int32), qw.astype(np.int32)
            phase("c4 open doc-range shard")
            ix = m.SparseIndex(path, device=ranks.local_rank, shard=ranks.rank, n_shards=ranks.world)
        except Exception as e:
            err = e
        if ranks.any_failed(err is not None):
            raise RuntimeError(f"opening the index shard failed on some rank (this rank: {err})")
        exchange = "one RCCL all-gather of per-shard top-k + exact merge"
        step = None
        fallback_result = {}
        comm_facts = None
        if sharded:
            phase("c4 RCCL communicator init (doc-range shards)")
            uid = ranks.bcast_bytes(m.comm_unique_id() if ranks.rank == 0 else None, 128)
            try:
                ix.comm_init(ranks.world, ranks.rank, uid)
                comm_facts = ix.comm_info()
            except Exception as e:
                err = e
            if ranks.any_failed(err is not None):
                msg = f"RCCL communicator init failed on some rank (this rank: {err})"
                if not args.allow_gloo_fallback:
                    status["exit"] = EXIT_COMM
                    raise RuntimeError(msg + "; pass --allow-gloo-fallback to move the lists through gloo instead")
                # explicit opt-in: keep the doc-range shards and move the per-shard lists through torch.distributed
                # (gloo) instead, merged by msr_merge_lists
                log(f"[bench r{ranks.rank}] {msg}; --allow-gloo-fallback: gloo all-gather of the lists")
                from mllm_sparse_retrieval_amd import dist as mdist

                exchange = f"FALLBACK: gloo all-gather of host lists + msr_merge_lists ({msg})"
                sharded = False  # batch.search without the in-library exchange

                def step():
                    batch.search(10)
                    o, _, su, cnt = batch.fetch()
                    g = mdist.all_gather_lists(ranks.dist, o, su, cnt)
                    fallback_result["r"] = ix.merge_lists(g[0], g[1], g[2], 10)
        batch = ix.batch(qp, qt, qw, 10)
        phase("c4 timed region (doc-range shards)" if ranks.world > 1 else "c4 timed region")
        wall, score_ms, merge_ms = timed_steps(batch, 10, args.steps, args.warmup, ranks, sharded=sharded, step=step)
        nq = len(qp) - 1
        out = {"workload": f"synthetic {args.c4_docs} docs x128 nnz ({ix.n_postings} postings), {nq} queries x120 nnz, "
                           f"V=30000, top-10",
               "value": round(nq * args.steps / wall, 1), "unit": "queries/s", "n_gpus": ranks.world,
               "ms_per_step": round(wall / args.steps * 1e3, 3), "scaling": "strong",
               "parallelism": "1 GPU" if ranks.world == 1 else
               f"index DOC-range sharded over {ranks.world} GPUs ({ix.shard_ntiles} of {ix.n_tiles} tiles on rank 0), "
               + exchange + " — the scaling path; term-range shards (below) are exact but exchange-bound"}
        resident = ranks.gather_objects(ix.resident_bytes)
        facts = ranks.gather_objects(comm_facts)
        if ranks.rank == 0:
            out["resident_index_bytes_per_rank"] = resident
            if ranks.world > 1:
                out["rccl_ranks"] = facts[0][0] if facts[0] else None
                out["rccl_comm"] = [{"ranks": f[0], "rank": f[1], "device": f[2]} if f else None for f in facts]
            out["roofline"] = sparse_roofline(batch, 10, score_ms, "c4_1m" if ranks.world == 1 else f"c4_1m_shard{ranks.world}")
            out["roofline"]["merge_and_exchange_ms"] = round(merge_ms, 4)
            if ranks.world == 1 and not args.no_cpu:
                phase("c4 cpu baseline")
                cb, par = cpu_baseline(wl, batch.fetch(), args.cpu_seconds, args.cpu_threads)
                out["cpu_baseline"] = cb
                out["parity"] = par
                out["speedup_vs_cpu"] = round(out["value"] / cb["value"], 1)
                out["speedup_vs_best_cpu_row"] = round(out["value"] / cb["best_one_call_value"], 1)
        doc_sharded_result = batch.fetch() if sharded else fallback_result.get("r")
        batch.close()
        if sharded:
            ix.comm_destroy()
        ix.close()
        if sharded and not args.no_term_shards:
            # the north star's literal partition: term-range shards, each rank RESIDENT with its own term range only
            # (msr_index_open_termshard). Exact, but the exchange is a reduce-scatter of u32 accumulators
            # (nq x N x 4 B), so it is exchange-bound by construction.
            terr = None
            ixf = tb = None
            try:
                phase("c4 open term-range shard")
                ixf = m.SparseIndex(path, device=ranks.local_rank, term_shard=(ranks.rank, ranks.world))
                uid = ranks.bcast_bytes(m.comm_unique_id() if ranks.rank == 0 else None, 128)
                phase("c4 RCCL communicator init (term-range shards)")
                ixf.comm_init(ranks.world, ranks.rank, uid)
                tb = ixf.batch(qp, qt, qw, 10)
            except Exception as e:
                terr = e
            if ranks.any_failed(terr is not None):
                status["exit"] = EXIT_COMM
                out["term_range_shards"] = {"error": f"setup failed on some rank (this rank: {terr})"}
            else:
                phase("c4 timed region (term-range shards)")
                tsteps = max(1, min(args.steps, 3))
                twall, _, _ = timed_steps(tb, 10, tsteps, 1, ranks, sharded="terms")
                tres = tb.fetch()
                same = all((x == y).all() for x, y in zip(tres, doc_sharded_result))
                tres_bytes = ranks.gather_objects(ixf.resident_bytes)
                out["term_range_shards"] = {
                    "value": round(nq * tsteps / twall, 1), "unit": "queries/s", "ms_per_step": round(twall / tsteps * 1e3, 2),
                    "steps": tsteps, "identical_to_doc_range_result": bool(same),
                    "resident_index_bytes_per_rank": tres_bytes,
                    "term_range_rank0": [ixf.term_lo, ixf.term_hi],
                    "exchange": "ncclReduceScatter(sum) of u32 accumulator tiles, then ncclAllGather of per-range top-k"}
                if not same:
                    status["exit"] = 1
            if tb is not None:
                tb.close()
            if ixf is not None:
                ixf.comm_destroy()
                ixf.close()
    finally:
        ranks.barrier()
        if ranks.rank == 0:
            try:
                os.remove(path)
            except OSError:
                pass
    return out


def c5_parity_sample(m, docs, n_terms, qp, qt, qw, p, q, depth, k, alpha, got, every, remove_query=False, tie_key=None):
    """GPU hybrid results of every `every`-th query against the ORACLE pipeline: C oracle sparse top-depth + numpy
    dense top-depth on the fp16-rounded inputs -> oracle.get_run_dict -> oracle.fuse (pinned to src/hybrid.py:32-53)
    -> top-k. Fused scores within 1e-5; ids equal except near-ties (2e-6). Query i carries the id str(i) (= a doc id:
    remove_query skips that doc, src/search.py:72-74); tie_key = the order that cuts exact dense ties (doc ordinals)."""
    from oracle import oracle, taat

    ords, fs, cnt, docid_of = got
    n, nq = p.shape[0], q.shape[0]
    sample = np.arange(0, nq, every)
    ids = [str(i) for i in range(n)]
    oix, order = taat.TaatIndex.from_rows_by_docid(*docs, n_terms, ids)
    sorted_ids = [ids[r] for r in order]
    sel = np.concatenate([np.arange(qp[i], qp[i + 1]) for i in sample])
    sp = np.concatenate([[0], np.cumsum(qp[sample + 1] - qp[sample])]).astype(np.int64)
    wo, wsc, wn = oix.search(sp, qt[sel], qw[sel], depth, threads=16)
    s = q[sample].astype(np.float16).astype(np.float32) @ p.astype(np.float16).astype(np.float32).T
    tk = np.arange(n) if tie_key is None else np.asarray(tie_key, dtype=np.int64)
    didx = np.lexsort((np.broadcast_to(tk, s.shape), -s), axis=1)[:, :depth]
    dsc = np.take_along_axis(s, didx, axis=1)
    qids = [str(int(i)) for i in sample]
    o_sparse = oracle.get_run_dict(qids, [[float(np.float32(x)) for x in wsc[j, :wn[j]]] for j in range(len(sample))],
                                   [[sorted_ids[int(d)] for d in wo[j, :wn[j]]] for j in range(len(sample))], remove_query)
    o_dense = oracle.get_run_dict(qids, dsc, np.array([[ids[j] for j in row] for row in didx]), remove_query)
    want = oracle.fuse([o_dense, o_sparse], [alpha, 1 - alpha])
    worst, id_mism, near_tie_swaps = 0.0, 0, 0
    for j, i in enumerate(sample):
        ranked = sorted(want[qids[j]].items(), key=lambda kv: (-float(kv[1]), kv[0].encode()))[:k]
        if cnt[i] != len(ranked):
            id_mism += 1
            continue
        for r, (doc, score) in enumerate(ranked):
            worst = max(worst, abs(float(fs[i, r]) - float(score)))
            g = docid_of(int(ords[i, r]))
            if g != doc:
                if g in want[qids[j]] and abs(float(want[qids[j]][g]) - float(score)) <= 2e-6:
                    near_tie_swaps += 1
                else:
                    id_mism += 1
    return {"checked_queries": int(len(sample)), "max_abs_score_diff": worst, "score_tolerance": 1e-5,
            "id_mismatches": id_mism, "near_tie_swaps": near_tie_swaps,
            "checker": "oracle pipeline: oracle_taat.c + numpy dense on fp16-rounded inputs + oracle.fuse"}


def run_c5(args, ranks, m, wlmod, shape="t2i"):
    """Hybrid dense + sparse search, H = 4096, depth 1000 -> fused top-10, alpha 0.5 (scripts/search.sh:25,32). One GPU.
    shape "t2i" = BASELINE configs[4] (COCO-5K text->image): 5 000 image docs, 25 010 caption queries: one tile, one
        fused kernel per query (hybrid_tiles).
    shape "i2t" = the run the reference's own script does (scripts/search.sh:5,26-27: TARGET_TYPE=text, --query_type
        image, --remove_query): 25 010 caption docs, 5 000 image queries: four tiles, the candidate kernels
        (hybrid_tiles<MODE 1> per (tile, query) + hybrid_fuse_query per query)."""
    from mllm_sparse_retrieval_amd.dense import DenseIndex, hybrid_search, row_to_ordinal

    h, depth, k, alpha, n_terms = 4096, 1000, 10, 0.5, 30000
    n, nq = (args.c5_docs, args.c5_queries) if shape == "t2i" else (args.c5_queries, args.c5_docs)
    remove = shape == "i2t"
    wname = "c5_hybrid" if shape == "t2i" else "c5_hybrid_i2t"
    docs, (qp, qt, qw), p, q = wlmod.hybrid_vectors(n, nq, h, n_terms, seed=4, threads=args.host_threads)
    tmp = tempfile.mkdtemp(prefix="msr_c5_")
    path = m.build_index_from_csr(os.path.join(tmp, "c5.idx"), *docs, n_terms, threads=args.host_threads)
    ix = m.SparseIndex(path, device=ranks.local_rank)
    dix = DenseIndex(p, device=ranks.local_rank)
    r2o = row_to_ordinal(ix, [str(i) for i in range(n)])
    self_ord = r2o[:nq].astype(np.int32) if remove else None   # query i carries the id of doc i (remove_query)
    phase(f"{wname} hybrid search")
    hybrid_search(ix, dix, qp, qt, qw, q, depth, k, alpha, r2o, self_ord)                      # warm-up
    t0 = time.perf_counter()
    ords, fs, cnt, ms = hybrid_search(ix, dix, qp, qt, qw, q, depth, k, alpha, r2o, self_ord)
    wall = time.perf_counter() - t0
    kern = sum(ms.values())
    flops = 2.0 * nq * n * h
    fused = ms["dense_select"] == 0 and ms["fusion"] == 0     # one tile: hybrid_tiles does everything but the GEMM
    multi = ms["dense_select"] == 0 and ms["fusion"] > 0      # several tiles: candidate kernel + per-query fusion kernel
    if fused:
        kernel_ms = {"hybrid_tiles": round(ms["sparse"], 3), "dense_gemm": round(ms["dense_gemm"], 3)}
        pipeline = ("dense_scores_256k (fp16 MFMA GEMM, 16x16x32, four waves of 128x128, on the ordinal-ordered passage "
                    "matrix, query chunks of ~160 MB of score rows) -> hybrid_tiles (one workgroup per query: sparse "
                    f"scores in LDS, both depth-{depth} memberships by one histogram pass, min-max fusion, top-{k})")
    elif multi:
        kernel_ms = {"hybrid_tiles_mode1": round(ms["sparse"], 3), "dense_gemm": round(ms["dense_gemm"], 3),
                     "hybrid_fuse_query": round(ms["fusion"], 3)}
        pipeline = ("dense_scores_256k (as above) -> hybrid_tiles<MODE 1> (one workgroup per (tile, query): sparse scores "
                    "in LDS, the tile's quota of candidates for both depth lists by one histogram pass) -> "
                    f"hybrid_fuse_query (one workgroup per query: both depth-{depth}-th bests among the candidates by "
                    f"radix select, verified against every tile's weakest candidate, fusion in an LDS hash table, top-{k}); "
                    "flagged queries (none here) are repeated with quota = depth")
    else:
        kernel_ms = {k2: round(v, 3) for k2, v in ms.items()}
        pipeline = "list-based: score_tiles + dense GEMM + select_tiles + fuse_tiles + merges"
    out = {"workload": f"hybrid: {n} docs x (128 nnz + {h}-d fp16), {nq} queries x (120 nnz + {h}-d), depth {depth} -> "
                       f"fused top-{k}, alpha {alpha}" + (", remove_query" if remove else ""),
           "value": round(nq / (kern * 1e-3), 1), "unit": "queries/s (kernel time, inputs resident)",
           "host_inclusive_queries_per_s": round(nq / wall, 1),
           "kernel_ms": kernel_ms, "n_tiles": ix.n_tiles,
           "pipeline": pipeline,
           "dense_tflops": round(flops / (ms["dense_gemm"] * 1e-3) / 1e12, 1) if ms["dense_gemm"] > 0 else None,
           "dtype": "f16 in / f32 accumulate (dense), u32 (sparse), f32 (fusion)"}
    # roofline per stage: the GEMM against the dense fp16 MFMA peak; the integer / selection kernels against the
    # ceilings they can run into (counter-backed where profiled)
    rl = {}
    if ms["dense_gemm"] > 0:
        tf = flops / (ms["dense_gemm"] * 1e-3) / 1e12
        rl["dense_gemm"] = {"bound": "mfma", "achieved": round(tf, 1), "peak": MFMA_F16_PEAK_TF, "unit": "TFLOP/s",
                            "frac": round(tf / MFMA_F16_PEAK_TF, 4), "kernel_ms": round(ms["dense_gemm"], 4),
                            "utilisation": binding_fractions(counters(wname, "dense_scores"), ms["dense_gemm"])}
    stages = ((("hybrid_tiles", "hybrid_tiles", "sparse"),) if fused else
              (("hybrid_tiles_mode1", "hybrid_tiles", "sparse"), ("hybrid_fuse_query", "hybrid_fuse_query", "fusion")) if multi else
              (("dense_select", "select_tiles", "dense_select"), ("fusion", "fuse_tiles", "fusion"),
               ("sparse", "score_tiles", "sparse")))
    # the sparse work of one search (the hybrid kernels run the search path's accumulation): postings by pipe, accumulator tiles
    wb = ix.batch(qp, qt, qw, 1)
    work = wb.work()
    wb.close()
    pk = pipe_peaks()
    for stage, kre, key in stages:
        if ms.get(key, 0) > 0:
            fr = binding_fractions(counters(wname, kre), ms[key])
            cands = {b: fr.get(k3) for b, k3 in (("hbm", "hbm_frac"), ("l2", "l2_frac"), ("valu", "valu_busy"))
                     if fr.get(k3) is not None}
            busiest = max(cands, key=cands.get) if cands else None
            rl[stage] = {"bound": busiest, "frac": None, "busiest_pipe_utilisation": cands.get(busiest) if busiest else None,
                         "kernel_ms": round(ms[key], 4), "utilisation": fr}
            # useful-work fraction as for score_tiles (sparse_roofline), for the kernels that score: the accumulator tile
            # is written twice (init, fused keys) and read four times (two selection passes, fusion, final collection)
            # by hybrid_tiles, written once and read three times (two selection passes, emission) by its MODE 1
            passes = {"hybrid_tiles": 6, "hybrid_tiles_mode1": 4}.get(stage)
            if passes and pk.get("ds_add_per_s"):
                min_ms = {"lds_atomic": work["sparse_postings"] / pk["ds_add_per_s"] * 1e3,
                          "valu_dot2": work["dense_head_postings"] / 2.0 / pk["dot2_per_s"] * 1e3,
                          "lds_bw": work["acc_init_bytes"] * passes / pk["lds_bytes_per_s"] * 1e3}
                if fr.get("traffic") is not None:
                    min_ms["hbm"] = fr["traffic"] / (HBM_PEAK_GBS * 1e9) * 1e3
                bound = max(min_ms, key=min_ms.get)
                complete = "hbm" in min_ms and not counters_state()["stale"]
                rl[stage].update(bound=bound, frac=round(min_ms[bound] / ms[key], 4) if complete else None,
                                 useful={"min_ms_per_pipe": {b: round(v, 4) for b, v in min_ms.items()},
                                         "accumulator_passes": passes, "work_per_search": work,
                                         "note": "binding pipe's minimum time at the measured instruction peaks / kernel time; "
                                                 "the two depth-1000 selections and the fusion are this kernel's purpose but "
                                                 "not 'work' in this sense: only their accumulator traffic is counted"})
    out["roofline"] = rl
    out["counters_stale"] = bool(counters_state()["stale"])
    if not args.no_cpu and shape == "i2t":
        # the dense search alone at the batch sizes the reference runs it with (PCIe-inclusive: host f32 queries in, host
        # lists out): --batch_size 2 (scripts/search.sh:29) and the default 128 (src/arguments.py:60)
        rows = {}
        for bs in (2, 128):
            dix.search(q[:bs], depth)
            a0 = dix.stats()["device_allocs"]
            t0 = time.perf_counter()
            for _ in range(20):
                dix.search(q[:bs], depth)
            dt = (time.perf_counter() - t0) / 20
            rows[f"batch_{bs}"] = {"ms_per_call": round(dt * 1e3, 4), "queries_per_s": round(bs / dt, 1),
                                   "kernel_ms": {"gemm": round(dix.last_ms[0], 4), "select": round(dix.last_ms[1], 4)},
                                   "device_allocs_in_20_calls": dix.stats()["device_allocs"] - a0}
        out["host_inclusive_dense_search"] = rows
    if not args.no_cpu:
        phase(f"{wname} parity sample vs the oracle pipeline")
        out["parity"] = c5_parity_sample(m, docs, n_terms, qp, qt, qw, p, q, depth, k, alpha,
                                         (ords, fs, cnt, ix.docid), every=50 if shape == "t2i" else 25,
                                         remove_query=remove, tie_key=r2o)
    dix.close()
    ix.close()
    try:
        os.remove(path)
        os.rmdir(tmp)
    except OSError:
        pass
    return out


def main():
    argv = sys.argv[1:]
    args = parse_args(argv)
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # nothing has touched the GPU yet (torch.cuda.device_count() does not): start the ranks as children
        sys.exit(launch_ranks(args, argv))

    # stdout carries exactly ONE JSON line: keep a private handle to it and point fd 1 at stderr, so that banners
    # printed by native libraries (RCCL prints its version block at communicator init) cannot pollute it
    real_stdout = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    global _SYNC_DEVICE, _M
    if args.launcher_selftest:
        ranks = Ranks(args)
        ranks.barrier()
        top = ranks.max(float(ranks.rank))
        if ranks.rank == 0:
            print(json.dumps({"launcher_selftest": True, "n_gpus": ranks.world, "max_rank_seen": int(top)}),
                  file=real_stdout, flush=True)
        ranks.close()
        return

    import mllm_sparse_retrieval_amd as m  # raises if libmsr.so is missing: there is no fallback scorer
    from mllm_sparse_retrieval_amd import _cabi
    from mllm_sparse_retrieval_amd import workloads as wlmod

    _cabi.lib()   # libmsr.so (and with it /opt/rocm's HIP runtime + librccl) BEFORE torch, see the module docstring
    _M = m
    ranks = Ranks(args)
    _SYNC_DEVICE = ranks.local_rank

    if args.dense_max >= 0:
        m.set_build_option("dense_max_terms", args.dense_max)
    if args.dense_density >= 0:
        m.set_build_option("dense_min_density", args.dense_density)

    only = args.only_c3 or args.only_c4 or args.only_c5
    out = {}
    status = {"exit": 0}
    if not only:
        out = run_headline(args, ranks, m, wlmod)
    if ranks.world == 1 and not args.no_c3 and (args.only_c3 or not only):
        try:
            out["c3_coco5k"] = run_c3(args, ranks, m, wlmod)
        except Exception as e:
            out["c3_coco5k"] = {"error": f"{type(e).__name__}: {e}"}
            log(f"[bench] c3_coco5k failed: {out['c3_coco5k']['error']}")
    if ranks.world == 1 and not args.no_c5 and (args.only_c5 or not only):
        for name, shape in (("c5_hybrid", "t2i"), ("c5_hybrid_i2t", "i2t")):
            if args.c5_shape not in ("both", shape):
                continue
            try:
                out[name] = run_c5(args, ranks, m, wlmod, shape)
            except Exception as e:
                out[name] = {"error": f"{type(e).__name__}: {e}"}
                log(f"[bench] {name} failed: {out[name]['error']}")
    hung = False
    if not args.no_c4 and (args.only_c4 or not only):
        # The extra object must never take the headline NUMBER down: exceptions are caught and recorded, and at N > 1 a
        # watchdog bounds the wait so that the JSON line is printed regardless — but a failure or a hang there still
        # ends the run with a non-zero status.
        box = {}

        def work():
            try:
                box["c4"] = run_c4(args, ranks, m, wlmod, status)
            except Exception as e:
                box["c4"] = {"error": f"{type(e).__name__}: {e}"}
                if ranks.world > 1 and status["exit"] == 0:
                    status["exit"] = 1
                log(f"[bench r{ranks.rank}] c4_1m failed: {box['c4']['error']}")

        if ranks.world > 1:
            import threading

            t = threading.Thread(target=work, daemon=True)
            t.start()
            t.join(args.c4_timeout)
            if t.is_alive():
                hung = True
                box["c4"] = {"error": f"no completion within {args.c4_timeout:.0f}s; rank {ranks.rank} was in phase "
                                      f"'{_PHASE}' (every rank logs its phases on stderr)", "hung_phase": _PHASE}
                log(f"[bench r{ranks.rank}] WATCHDOG: stuck in phase '{_PHASE}'")
        else:
            work()
        out["c4_1m"] = box["c4"]
    if hung:
        out["hung"] = True
    if status["exit"]:
        out["exit_status"] = status["exit"]
    if ranks.rank == 0:
        print(json.dumps(out), file=real_stdout, flush=True)
    if hung:
        real_stdout.flush()
        os._exit(EXIT_HUNG)  # a native call is stuck: leave without joining it, and say so with the status
    ranks.close()
    if status["exit"]:
        sys.exit(status["exit"])


if __name__ == "__main__":
    main()
