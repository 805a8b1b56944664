#!/usr/bin/env python3
"""bench.py - headline benchmark of the MI355X k-mer hot path.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME]

N > 1: one rank per GPU over RCCL - either launched by the driver through torch.distributed.run, or, typed as it stands
(`python bench.py --gpus 8`), bench.py starts torch.distributed.run itself as a child process (launch_ranks).
A "step" is one pass of the hot path over one batch of synthetic reads that is already
resident in HBM.

Default (no --workload) = BASELINE.json's whole headline metric, one after the other:
  * `comp oligo k=4`, 10 M x 150 bp reads per GPU, f64 rows (configs[1]) - the line's top-level
    `value` / `roofline` / `cpu_baseline`;
  * `ctr k=31`, 25 M x 150 bp reads per GPU (at --gpus 8 that is configs[3]) - the `ctr_k31` object with
    its own `value`, `ms_per_step`, `roofline`, `cpu_baseline` and the genome-sampled read distribution.
  * `ctr k=15` (configs[2]) and `comp cgr k=7` f32 (configs[4]) as the `ctr_k15` / `comp_cgr_k7` objects, each
    with its own `roofline`.
All run W warm-up and exactly K timed steps between barrier + synchronize brackets (max over ranks).

Self-certification: the line's `kt_env` lists every KT_* variable the process saw; a variable that can switch
work off (ablation builds only) makes bench.py exit non-zero before anything is timed.  After each timed loop,
untimed, the output is checked (`output_check`): oligo rows sum to 1 over the whole output and a 4096-row slice
is compared with the CPU oracle inside the cpu_baseline leg; ctr's exported counts sum to reads x (L - k + 1), and the
k-mers of a 20 000-read sample of rank 0's batch, counted by the CPU oracle, are looked up in the shards of all ranks
(`output_check.sampled_oracle`: each found on exactly one rank - its owner - with at least the sample's count).

Array placement (oligo / cgr workloads, untimed, before the ramp): where an array lies in the HBM moves the store-bound
kernels by up to 20 %, so the output array - and for one-batch workloads the input array - is the
fastest of up to eight candidate allocations, chosen BY THE LIBRARY (kt_device_alloc_placed in include/kmertools_hip.h:
any caller that links libkmertools_hip.so gets the same array).  The line's `output_placement` / `input_placement`
objects list every candidate's time and which was kept, candidate 0 being the plain allocation, and
`roofline.frac_plain_allocation` is the fraction a plain allocation would have given (candidate 0's probe time);
`--no-place` takes the plain allocation.  Nothing about the timed steps changes: same kernel, same work, same checks.

ctr's step is what SURVEY.md 8d puts inside it: clear + insert (at N > 1: route pass + exchange of the records + insert) +
kt_ctr_size + kt_ctr_export into device arrays; algorithmic bytes = L + kmers*16 per read + distinct*12.

`roofline` is for the step's kernels: algorithmic bytes per launch (DESIGN.md "Measurement") / the
average launch duration measured with HIP events on the launch stream inside the timed region.
`cpu_baseline` is the CPU oracle (a C restatement of the reference's algorithm - the Rust reference
cannot be built here) timed on this box's host cores on a bounded sample, rank 0, N = 1 only.
"""
import argparse
import json
import os
import pathlib
import statistics
import sys
import tempfile
import time

ROOT = pathlib.Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
SEED = 0x6b6d6572
METRIC = "Gbases/s on comp-oligo k=4 and ctr k=31, 150bp synthetic reads, 1/2/4/8 GPU"

WORKLOADS = {
    # name: kind, k, reads per GPU, read length, dtype, batch (reads per launch)
    "comp_oligo_k4": dict(kind="oligo", k=4, n=10_000_000, L=150, dtype="f64", batch=10_000_000, cfg=1,
                          desc="comp oligo k=4 canonical, 10M x 150bp synthetic reads per GPU, f64 rows"),
    "comp_cgr_k7": dict(kind="oligo", k=7, n=10_000_000, L=150, dtype="f32", batch=1_000_000, cfg=4,
                        desc="comp cgr k=7 (8192 canonical bins), 10M x 150bp per GPU, f32 rows, 1M-read output ring"),
    "ctr_k15": dict(kind="ctr", k=15, n=50_000_000, L=150, cfg=2,
                    desc="ctr k=15 canonical counts, 50M x 150bp per GPU, hash table in HBM, exported to device arrays"),
    "ctr_k31": dict(kind="ctr", k=31, n=25_000_000, L=150, cfg=3,
                    desc="ctr k=31 canonical counts, 25M x 150bp per GPU (200M over 8), hash-prefix sharded, "
                         "exported to device arrays"),
    # next row of SURVEY.md 8f: per-read coverage histograms against the resident table (table build untimed)
    "cov_k15": dict(kind="cov", k=15, n=10_000_000, L=150, dtype="f64", cfg=5, bin_size=16, bin_count=16,
                    desc="cov k=15 bin_size 16 x 16 bins, 10M x 150bp per GPU against the table of the same reads, f64 rows"),
    # whole-sequence chaos game walk, 16 bytes of f64 points per base
    "cgr_whole": dict(kind="cgr", k=0, n=10_000_000, L=150, dtype="f64", cfg=6,
                      desc="comp cgr (no -k) whole-sequence walk, vecsize 1, 10M x 150bp per GPU, (x,y) f64 per base"),
    # window minimisers (w=31, m=7), variable-length (mmer, start, end) lists per read
    "min_w31_m7": dict(kind="min", k=7, n=10_000_000, L=150, dtype="u64", cfg=7, w=31,
                       desc="min w=31 m=7, 10M x 150bp per GPU, (minimiser, start, end) triples in read order"),
}
HEADLINE = ("comp_oligo_k4", "ctr_k31", "ctr_k15", "comp_cgr_k7")
# environment variables that (in an ablation build of the library, kmertools_amd/variants/) switch phases of the
# timed kernels off; the default build ignores them, the benchmark refuses to run with them
WORK_SKIPPING_ENV = ("KT_OLIGO_DEBUG", "KT_BUILD_DBG")
CHECK_ROWS = 4096
GENOME_LEN = 1_000_000_000   # second ctr distribution, reads sampled from a random 1 Gbp genome


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--no-place", action="store_true",
                    help="oligo workloads: take the first output allocation instead of the fastest of up to eight")
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="headline", choices=["headline"] + sorted(WORKLOADS))
    ap.add_argument("--reads", type=int, default=0, help="override reads per GPU (debug; marks the line as reduced)")
    ap.add_argument("--genome", type=int, default=0, help="ctr: sample reads from a random genome of this length")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-genome", action="store_true", help="headline: skip ctr's genome-sampled distribution")
    ap.add_argument("--no-export", action="store_true", help="ctr: time clear + insert only (round-1 region; experiments)")
    ap.add_argument("--cap-log2", type=int, default=0, help="ctr: override log2 of the table capacity (experiments)")
    ap.add_argument("--cap-slots", type=float, default=0, help="ctr: override the requested table capacity (experiments)")
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    return ap.parse_args()


def kt_environment():
    """every KT_* variable this process sees (recorded in the line); exits on a work-skipping one"""
    seen = {k: v for k, v in sorted(os.environ.items()) if k.startswith("KT_")}
    bad = [k for k in WORK_SKIPPING_ENV if os.environ.get(k, "") not in ("", "0")]
    if bad:
        sys.exit("bench.py: %s set - these switch work off in ablation builds of the library; refusing to time "
                 "anything (unset them)" % ", ".join(bad))
    return seen


def effective_cores():
    """host threads this process may really use: os.cpu_count() capped by the cgroup CPU quota"""
    n = os.cpu_count() or 1
    try:
        q, p = pathlib.Path("/sys/fs/cgroup/cpu.max").read_text().split()[:2]
        if q != "max":
            n = min(n, max(1, -(-int(q) // int(p))))
    except (OSError, ValueError):
        pass
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    return n


# ---- cpu_baseline legs (the only place bench.py touches oracle/) ------------------------------------------

def cpu_baseline_oligo(k, L, seconds):
    """CPU oracle (restatement of composition/src/oligo.rs:231-259 + the rayon par_iter of
    pybindings/src/oligo.rs:77-81) on the host cores, and on one core."""
    from oracle import kt_oracle as oracle
    cores = effective_cores()
    n2 = 2_000_000
    hb, ho = oracle.synth_reads(SEED, n2, L)
    oracle.oligo_batch(hb[:100_000 * L], ho[:100_001], k, True, True, 1.0, threads=cores)   # warm
    reps, t0 = 0, time.perf_counter()
    while True:   # bounded sample: repeat the 2 M-read batch until ~`seconds` of CPU work
        oracle.oligo_batch(hb, ho, k, True, True, 1.0, threads=cores)
        reps += 1
        dt = time.perf_counter() - t0
        if dt >= seconds or reps >= 200:
            break
    n1 = 400_000
    t1 = time.perf_counter()
    oracle.oligo_batch(hb[:n1 * L], ho[:n1 + 1], k, True, True, 1.0, threads=1)
    dt1 = time.perf_counter() - t1
    return dict(value=reps * n2 * L / dt / 1e9, unit="Gbases/s", cores=cores, kind="port", label="restatement",
                value_1_thread=n1 * L / dt1 / 1e9,
                sample="%d passes over %d x %dbp synthetic reads, k=%d canonical f64 rows, %d threads, %.1f s; "
                       "1 thread: %d reads, %.1f s" % (reps, n2, L, k, cores, dt, n1, dt1))


def cpu_baseline_ctr(k, L, seconds, genome):
    """CPU oracle (restatement of counter/src/lib.rs:100-131: n_parts concurrent maps locked per bucket like scc's;
    :151-167 text spill; :188-231 merge) on the host cores: in memory at T = cores, 4 and 1 - every run into maps
    sized for the same 1 M-read sample, so that the three figures differ by the threads alone - and with the
    temp-file round trip the real CLI pays."""
    from oracle import kt_oracle as oracle
    cores = effective_cores()
    parts = max(cores, 1)          # counter/src/lib.rs:243-247: n_parts = max(threads, ...)
    n = 1_000_000                  # at least 1 M reads
    kmers = n * (L - k + 1)
    hb, ho = oracle.synth_reads(SEED, n, L, genome_len=genome)

    def timed(threads, reads):
        c = oracle.Counter(parts)
        c.reserve(kmers)           # (scc grows while it adds; here the maps get their final size up front)
        t0 = time.perf_counter()
        c.add_reads(hb[:reads * L], ho[:reads + 1], k, threads=threads)
        dt = time.perf_counter() - t0
        del c
        return reads * L / dt / 1e9, dt
    v_all, dt_all = timed(cores, n)
    # fewer threads count a prefix of the sample (bounded time), into maps of the same size
    n4 = min(n, max(n * 4 // max(cores, 4), 50_000))
    v4, dt4 = timed(min(4, cores), n4)
    n1 = min(n, max(n // max(cores, 1), 50_000))
    v1, dt1 = timed(1, n1)
    # a smaller chunk through count_chunk's spill and merge's re-parse as well (what `kmertools ctr` does with
    # every chunk: counter/src/lib.rs:151-167, :195-216)
    n3 = 120_000
    c = oracle.Counter(parts)
    with tempfile.TemporaryDirectory(prefix="kt_cpu_ctr_") as d:
        t0 = time.perf_counter()
        c.add_reads(hb[:n3 * L], ho[:n3 + 1], k, threads=cores)
        c.spill(d, 0, threads=cores)
        lines = oracle.merge_files(d, parts, 1, threads=cores)
        dt_disk = time.perf_counter() - t0
    del c
    return dict(value=v_all, unit="Gbases/s", cores=cores, kind="port", label="restatement",
                value_4_threads=v4, value_1_thread=v1, scaling_vs_1_thread=round(v_all / v1, 2),
                value_with_text_spill_and_merge=n3 * L / dt_disk / 1e9,
                sample="%d x %dbp synthetic reads%s, k=%d, %d maps locked per 8-slot bucket, sized for the sample: %d "
                       "threads %.1f s; 4 threads: %d reads, %.1f s; 1 thread: %d reads, %.1f s; count + text spill + "
                       "merge: %d reads, %d lines, %.1f s"
                       % (n, L, " (genome-sampled)" if genome else "", k, parts, cores, dt_all, n4, dt4, n1, dt1, n3,
                          lines, dt_disk),
                note="the maps are sized before counting where scc grows while it adds, so the figure may over-state "
                     "the Rust reference a little; it does not under-state it")


def _threaded_passes(make_slice_fn, n, L, seconds, what):
    """Runs `fn()` for one slice of the reads per thread (the C oracle releases the GIL inside ctypes calls): the
    analogue of the reference's rayon par_iter over records.  Returns the cpu_baseline dict."""
    from concurrent.futures import ThreadPoolExecutor
    cores = effective_cores()
    per = (n + cores - 1) // cores
    fns = [make_slice_fn(i * per, min(n, (i + 1) * per)) for i in range(cores) if i * per < n]
    with ThreadPoolExecutor(len(fns)) as pool:
        list(pool.map(lambda f: f(), fns))   # untimed: first touch of the result buffers
        reps, t0 = 0, time.perf_counter()
        while True:
            list(pool.map(lambda f: f(), fns))
            reps += 1
            dt = time.perf_counter() - t0
            if dt >= seconds or reps >= 2000:
                break
    return dict(value=reps * n * L / dt / 1e9, unit="Gbases/s", cores=len(fns), kind="port", label="restatement",
                sample="%d passes over %d x %dbp synthetic reads, %s, %d threads, %.1f s" % (reps, n, L, what, len(fns), dt))


def cpu_baseline_cgr(L, seconds):
    """CPU oracle (restatement of composition/src/cgr.rs:127-144), reads split over the host cores."""
    import numpy as np
    from oracle import kt_oracle as oracle
    n = 800_000
    hb, ho = oracle.synth_reads(SEED, n, L)

    def make(lo, hi):
        b, o = hb[lo * L:hi * L], (ho[lo:hi + 1] - ho[lo]).copy()
        res = np.zeros(((hi - lo) * L, 2), np.float64)   # written in place by every pass
        return lambda: oracle.cgr_batch(b, o, 1, out=res)
    return _threaded_passes(make, n, L, seconds, "vecsize 1, result buffers reused")


def cpu_baseline_min(L, w, m, seconds):
    """CPU oracle (restatement of kmer/src/minimiser.rs:61-175), reads split over the host cores."""
    from oracle import kt_oracle as oracle
    n = 1_600_000
    hb, ho = oracle.synth_reads(SEED, n, L)

    def make(lo, hi):
        b, o = hb[lo * L:hi * L], (ho[lo:hi + 1] - ho[lo]).copy()
        return lambda: oracle.minimisers_batch_count(b, o, w, m)
    return _threaded_passes(make, n, L, seconds, "w=%d m=%d" % (w, m))


def cpu_baseline_cov(k, L, seconds, genome, bin_size, bin_count):
    """CPU oracle (restatement of coverage/src/lib.rs:165-184), table prebuilt, reads split over the host cores."""
    from oracle import kt_oracle as oracle
    n = 800_000
    hb, ho = oracle.synth_reads(SEED, n, L, genome_len=genome)
    c = oracle.Counter(1)
    c.add_reads(hb, ho, k)

    def make(lo, hi):
        b, o = hb[lo * L:hi * L], (ho[lo:hi + 1] - ho[lo]).copy()
        return lambda: c.cov_batch(b, o, k, bin_size, bin_count, True)
    return _threaded_passes(make, n, L, seconds, "k=%d, lookups in a prebuilt table" % k)


def cpu_baseline_for(wl, args, genome):
    k, L = wl["k"], wl["L"]
    if wl["kind"] == "oligo":
        return cpu_baseline_oligo(k, L, args.cpu_seconds)
    if wl["kind"] == "cgr":
        return cpu_baseline_cgr(L, args.cpu_seconds)
    if wl["kind"] == "min":
        return cpu_baseline_min(L, wl["w"], k, args.cpu_seconds)
    if wl["kind"] == "cov":
        return cpu_baseline_cov(k, L, args.cpu_seconds, genome, wl["bin_size"], wl["bin_count"])
    return cpu_baseline_ctr(k, L, args.cpu_seconds, genome)


# ---- one workload: setup, W warm-up steps, exactly K timed steps -------------------------------------------

class Env:
    """what every workload run shares: torch, the process group, the library context"""

    def __init__(self, args):
        import torch
        import torch.distributed as dist
        from kmertools_amd import device
        self.torch, self.dist, self.device = torch, dist, device
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        if self.world != args.gpus:   # (launched by torch.distributed.run with another count: the launcher's is the truth)
            args.gpus = self.world
        # KT_BENCH_SHARE_GPU=1 (tests only): every rank uses GPU 0 and the collectives go over gloo, so the
        # multi-rank launch path can be exercised on a one-GPU box.  Numbers from such a run mean nothing.
        self.share_gpu = os.environ.get("KT_BENCH_SHARE_GPU") == "1"
        self.dev_index = 0 if self.share_gpu else local_rank
        torch.cuda.set_device(self.dev_index)
        if self.world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if self.share_gpu:
                dist.init_process_group("gloo")
            else:
                dist.init_process_group("nccl", device_id=torch.device("cuda", self.dev_index))
        self.stream = torch.cuda.current_stream()
        self.ctx = device.Context(self.dev_index, stream=self.stream.cuda_stream)

    def barrier(self):
        if self.world > 1:
            self.dist.barrier()
        self.torch.cuda.synchronize()

    def max_over_ranks(self, x):
        if self.world == 1:
            return x
        t = self.torch.tensor([x], dtype=self.torch.float64, device="cpu" if self.share_gpu else "cuda")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def close(self):
        self.ctx.close()
        if self.world > 1:
            self.dist.destroy_process_group()


def run_workload(env, name, args, genome=0, steps=None, warmup=None):
    """-> dict(value, ms_per_step, ms_median, ms_min, roofline, config, dtype, extra...) for one workload"""
    torch, device, ctx = env.torch, env.device, env.ctx
    steps = args.steps if steps is None else steps
    warmup = args.warmup if warmup is None else warmup
    wl = dict(WORKLOADS[name])
    reduced = False
    if args.reads:
        wl["n"] = args.reads
        wl["batch"] = min(wl.get("batch", args.reads), args.reads)
        reduced = True
    n, L, k = wl["n"], wl["L"], wl["k"]
    world, rank = env.world, env.rank

    bases = torch.empty(n * L, dtype=torch.uint8, device="cuda")
    offsets = torch.empty(n + 1, dtype=torch.int64, device="cuda")
    ctx.synth_reads(SEED + wl["cfg"], n, L, bases, offsets, genome_len=genome, first_read=rank * n)
    torch.cuda.synchronize()

    launches_per_step = 1
    extra = {}
    keep_alive = []   # library-owned arrays (kt_device_alloc_placed) the workload's tensors are views of
    finish = lambda: None
    alg_extra = lambda: 0
    if wl["kind"] == "oligo":
        B = wl["batch"]
        bins = device.bins(k, True)
        tdt = torch.float64 if wl["dtype"] == "f64" else torch.float32
        esz = 8 if wl["dtype"] == "f64" else 4
        nb = (n + B - 1) // B
        launches_per_step = nb
        # per-batch views of the CSR arrays (offsets are absolute; each batch passes its slice base)
        batch_args = []
        for b in range(nb):
            r0, r1 = b * B, min(n, (b + 1) * B)
            if b == 0:
                batch_args.append((bases, offsets, r1 - r0))
            else:
                # offsets slice must start at 0: rebase once, outside the timed region
                o = (offsets[r0:r1 + 1] - offsets[r0]).contiguous()
                batch_args.append((bases[r0 * L:], o, r1 - r0))
        alg_bytes_per_launch = min(B, n) * (L + bins * esz)
        # the kernel's name as rocprofv3 prints it (kt_oligo.hip: k <= 4 runs the dense variant, k = 7 the producer-wave
        # kernel, 5 and 6 the plain one; template arguments <K, canonical, KT_F64 0 / KT_F32 1 / KT_U32 2, 4 waves>)
        dominant = "%s<%d, true, %d, 4>" % ("oligo_sb_kernel_dense" if k <= 4 else "oligo_pw_kernel" if k >= 7 else "oligo_sb_kernel",
                                            k, {"f64": 0, "f32": 1, "u32": 2}[wl["dtype"]])
        parallelism = "reads sharded by rank, no data-path collective"

        # Where the output array lies decides how fast it can be written (the same kernel ran at 1.90-2.40 ms on 23
        # allocations of one process), so - untimed - up to eight candidate allocations are tried with the
        # first batch and the fastest is kept (`output_placement` in the line lists them all; candidate 0 is what a
        # plain allocation would have got; --no-place switches this off)
        out_bytes = B * bins * esz
        if args.no_place or env.share_gpu:
            out = torch.empty((B, bins), dtype=tdt, device="cuda")
        else:
            bb0, oo0, cnt0 = batch_args[0]
            # k = 4: a probe runs both launch shapes (a candidate is ranked by what it takes with either); afterwards the
            # library measures the shape for the array that was kept by itself (kt_oligo_tuning on)
            shapes = (32, 96) if k == 4 else (0,)

            def run_into(bases_addr, out_addr):
                for sh in shapes:
                    ctx.oligo_tuning(sh)
                    ctx.oligo(bases_addr, oo0, cnt0, k, out_addr, count_min=True, norm=True, total_step=1, dtype=wl["dtype"])
            out_arr, placed = ctx.alloc_placed(out_bytes, lambda addr: run_into(bb0, addr), candidates=8, launches=4)
            out = out_arr.tensor((B, bins), tdt)
            keep_alive.append(out_arr)
            extra["output_placement"] = placed
            if nb == 1:   # the same for the (much smaller) input: the reads are copied into every candidate
                def probe_in(addr):
                    device.view_tensor(addr, (bb0.numel(),), torch.uint8).copy_(bb0)
                    run_into(addr, out)
                in_arr, placed = ctx.alloc_placed(bb0.numel(), probe_in, candidates=8, launches=4)
                bases_p = in_arr.tensor((bb0.numel(),), torch.uint8)
                bases_p.copy_(bb0)
                keep_alive.append(in_arr)
                batch_args[0] = (bases_p, oo0, cnt0)
                extra["input_placement"] = placed
            del bb0
            ctx.oligo_tuning(True)

        def step():
            for (bb, oo, cnt) in batch_args:
                ctx.oligo(bb, oo, cnt, k, out, count_min=True, norm=True, total_step=1, dtype=wl["dtype"])
    elif wl["kind"] == "min":
        w = wl["w"]
        evo = torch.empty(n + 1, dtype=torch.int64, device="cuda")
        one = torch.empty(1, dtype=torch.int64, device="cuda")
        n_ev = ctx.minimisers(bases, offsets, n, w, k, evo, one, one, one, 0)   # sizing call, untimed
        mk = torch.empty(n_ev, dtype=torch.int64, device="cuda")
        ms = torch.empty(n_ev, dtype=torch.int64, device="cuda")
        me = torch.empty(n_ev, dtype=torch.int64, device="cuda")
        # bases read by the three passes that touch them + the triples and offsets written
        alg_bytes_per_launch = n * (L + 8) + n_ev * 24
        dominant = "min_tile_kernel count + emit passes (LDS sliding minimum), break scan, finalize; %d triples" % n_ev
        parallelism = "reads sharded by rank, no data-path collective"

        def step():
            ctx.minimisers(bases, offsets, n, w, k, evo, mk, ms, me, n_ev)
    elif wl["kind"] == "cgr":
        bad = torch.zeros(1, dtype=torch.int64, device="cuda")
        if args.no_place or env.share_gpu:
            out = torch.empty((n * L, 2), dtype=torch.float64, device="cuda")
        else:   # the 24 GB of points are a store stream like the oligo rows: the array is chosen the same way
            out_arr, extra["output_placement"] = ctx.alloc_placed(n * L * 16, lambda a: ctx.cgr(bases, offsets, n, 1, a, bad))
            out = out_arr.tensor((n * L, 2), torch.float64)
            keep_alive.append(out_arr)
        alg_bytes_per_launch = n * (L * 17 + 8)
        dominant = "cgr_kernel (128-base chunks per lane, bracketing start, LDS-transposed stores)"
        parallelism = "reads sharded by rank, no data-path collective"

        def step():
            ctx.cgr(bases, offsets, n, 1, out, bad)
    elif wl["kind"] == "cov":
        kmers_per_read = L - k + 1
        max_distinct = min(n * kmers_per_read, (4 ** k + 2 ** k) // 2)
        cap = 1 << max(20, (2 * max_distinct - 1).bit_length())
        table = device.Counter(ctx, k, cap)
        table.add_reads(bases, offsets, n)   # untimed: the step is the lookup pass
        bc = wl["bin_count"]
        out = torch.empty((n, bc), dtype=torch.float64, device="cuda")
        alg_bytes_per_launch = n * (L + kmers_per_read * 16 + bc * 8)
        dominant = "cov_kernel k=%d (one 16-byte table probe per k-mer)" % k
        parallelism = "reads sharded by rank, each rank probes the table of its own reads"
        extra["distinct_rank0"] = table.size()
        finish = table.close

        def step():
            table.cov(bases, offsets, n, wl["bin_size"], bc, out, norm=True, dtype="f64")
    else:
        from kmertools_amd import dist as ktdist
        kmers_per_read = L - k + 1
        # the most distinct keys this rank can end up owning (its share of the hash space; 3 % head room for the
        # imbalance between owners)
        max_distinct = min(int(n * kmers_per_read * (1.03 if world > 1 else 1.0)), (4 ** k + 2 ** k) // 2)
        # slots: 1.9x the most distinct keys (load factor ~0.5; measured sweep in DESIGN.md: the table image is what the
        # build writes and the export reads, but the LDS insert slows down steeply with the load); the library rounds
        # up to m * 2^j, m in 5..8 (1.9x, not 2x: the canonical 15-mers are 2^29 + 2^14, 2x is just past a power of two)
        # (N > 1: 1.5x - next to the exchange regions and the partition buffers a 1.9x shard does not fit 288 GB)
        cap = max(1 << 20, int(float(os.environ.get("KT_BENCH_CAP_FACTOR", "1.9" if world == 1 else "1.5")) * max_distinct))
        if args.cap_log2:
            cap = 1 << args.cap_log2
        if args.cap_slots:
            cap = int(args.cap_slots)
        counter = ktdist.ShardedCounter(ctx, k, cap, group=None if world == 1 else env.dist.group.WORLD,
                                        max_batch_bases=n * L)
        with_export = not args.no_export
        xk = torch.empty(max_distinct if with_export else 1, dtype=torch.int64, device="cuda")
        xc = torch.empty(max_distinct if with_export else 1, dtype=torch.int32, device="cuda")
        alg_bytes_per_launch = n * (L + kmers_per_read * 16)
        state = {"distinct": 0}
        fused = with_export and os.environ.get("KT_BENCH_EXPORT_TARGET", "1") != "0"
        if fused:
            # the export arrays are named before counting (kt_ctr_export_target): the range build writes its packed
            # (key, count) pairs straight into them and kt_ctr_export has nothing left to copy
            counter.table.export_target(xk, xc, max_distinct)
        kt_ = "unsigned int" if k <= 16 else "unsigned long"
        routed = world > 1 or os.environ.get("KT_SHARD_FORCE")
        packed = os.environ.get("KT_BULK_PACK", "1") != "0"
        dominant = ("ctr k=%d step: clear + %sbulk table build (scatter1y_kernel<%s, %s, 1024>, part2_swwc_kernel<%s>, "
                    "build_kernel<%s, ...>%s)%s"
                    % (k, "route_kernel + " if routed else "pack_segments_kernel + " if packed else "",
                       "RecordSource" if routed else "PackedSource" if packed else "ReadsSource", kt_, kt_, kt_,
                       " writing the export arrays" if fused else "",
                       " + size + export" + ("" if fused else " (dense_export_kernel)") if with_export else ""))
        parallelism = ("one table per GPU" if world == 1 and not os.environ.get("KT_SHARD_FORCE") else
                       "ownership by the hash prefix of the k-mer's minimiser: route pass (records of <= 8 k-mers as 2-bit bases, "
                       "one region per owner) -> exchange of the regions in pieces -> the single-GPU partition + range build over "
                       "the records; transport: " + counter.transport)
        finish = counter.close
        if with_export:
            alg_extra = lambda: state["distinct"] * 12

        def step():
            counter.clear()
            counter.add_reads(bases, offsets, n)   # N > 1: route + exchange + count (kt_sharded_add_reads)
            counter.finalize()
            if with_export:
                d = counter.size_local()
                got = counter.table.export(xk, xc, max_distinct)
                assert got == d
                state["distinct"] = d

    # Ramp: untimed launches ahead of the W warm-up steps.  The store-bound kernels reach their steady rate only after
    # ~20 launches (clocks / power state after the idle set-up phase): oligo k=4 measured 2.115 ms per step with
    # --steps 10 --warmup 3, 2.03 with 20 / 5, 2.00 with 100 / 20 and with 10 / 60, same box, same minute.  The count
    # is fixed per workload so that every rank of a collective step runs it the same number of times.
    ramp = 1 if wl["kind"] == "ctr" else 60
    # (the first 20 launches of the ramp are timed as well - `ms_first_20_unramped` in the line - so that what the ramp is
    # worth stays on record: VERDICT r3)
    rev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(min(20, ramp))]
    for i in range(ramp):
        if i < len(rev):
            rev[i][0].record(env.stream)
        step()
        if i < len(rev):
            rev[i][1].record(env.stream)
    for _ in range(warmup):
        step()
    env.barrier()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    t0 = time.perf_counter()
    for i in range(steps):
        ev[i][0].record(env.stream)
        step()
        ev[i][1].record(env.stream)
    env.barrier()
    elapsed = env.max_over_ranks(time.perf_counter() - t0)

    ms_per_step = elapsed / steps * 1e3
    value = world * n * L * steps / elapsed / 1e9
    per_step_ms = [a.elapsed_time(b) for a, b in ev]
    kern_ms = sum(per_step_ms) / (steps * launches_per_step)
    alg_bytes_per_launch += alg_extra()
    achieved = alg_bytes_per_launch / (kern_ms * 1e-3) / 1e9

    # HBM bytes per launch from PMC counters (collected in separate rocprofv3 --pmc passes and
    # committed under profiles/; see profiles/traffic.json) - null when not collected
    torch.cuda.synchronize()
    first_ms = [a.elapsed_time(b) for a, b in rev]
    traffic, traffic_file, traffic_head = None, None, None
    try:
        key = name + ("_genome" if genome else "")
        if wl["kind"] == "ctr" and world == 1 and os.environ.get("KT_SHARD_FORCE") == "8" and not genome:
            key = name + "_forced8"   # (the one-GPU run of the 8-rank path has a profile of its own)
        elif wl["kind"] == "ctr" and (world > 1 or os.environ.get("KT_SHARD_FORCE")):
            key = None                # (no committed counter profile of this configuration)
        tj = json.loads((ROOT / "profiles" / "traffic.json").read_text()).get(key) if key else None
        if tj and not reduced and not (wl["kind"] == "ctr" and args.no_export):
            traffic = int((tj["fetch_kib"] * tj.get("fetch_correction", 1) + tj["write_kib"]) * 1024)
            traffic_file, traffic_head = tj.get("source"), tj.get("profile_head")
    except (OSError, ValueError, KeyError):
        traffic = None

    # ---- untimed: is the output the timed steps produced the right one? ------------------------------------------
    check, check_slice = None, None
    if wl["kind"] == "oligo":
        # one more pass over every batch: every row must sum to 1 (synthetic reads have L - k + 1 k-mers each); the
        # first CHECK_ROWS rows go to the host for the oracle comparison made in the cpu_baseline leg
        tol = 1e-12 if wl["dtype"] == "f64" else 1e-6
        bad, rows = 0, 0
        for bi, (bb, oo, cnt) in enumerate(batch_args):
            ctx.oligo(bb, oo, cnt, k, out, count_min=True, norm=True, total_step=1, dtype=wl["dtype"])
            sums = out[:cnt].sum(dim=1, dtype=torch.float64)
            bad += int(((sums - 1.0).abs() > tol).sum().item())
            rows += cnt
            if bi == 0 and rank == 0:
                m = min(CHECK_ROWS, cnt)
                check_slice = dict(kind="oligo", k=k, dtype=wl["dtype"], rows=m,
                                   bases=bb[:int(oo[m].item())].cpu().numpy(), offsets=oo[:m + 1].cpu().numpy(),
                                   out=out[:m].cpu().numpy())
            del sums
        check = {"rows": rows, "rows_not_summing_to_1": bad, "tolerance": tol, "ok": bad == 0}
    elif wl["kind"] == "ctr" and not args.no_export:
        d = state["distinct"]
        tot = torch.stack([xc[:d].sum(dtype=torch.int64), torch.tensor(d, device="cuda")])
        if world > 1:
            tot = tot.cpu() if env.share_gpu else tot
            env.dist.all_reduce(tot)
        want = world * n * kmers_per_read
        check = {"sum_of_counts": int(tot[0].item()), "expected": want, "distinct_all_ranks": int(tot[1].item()),
                 "ok": int(tot[0].item()) == want}
        # ... and is it the right table?  Rank 0 counts a sample of ITS OWN batch with the CPU oracle (the checker, as in the
        # cpu_baseline leg - nothing here is timed); every rank looks the sample's k-mers up in its shard (kt_ctr_lookup):
        # each must be found on exactly one rank - the rank kt_shard_owner_of gives it - with at least the sample's count.
        check["sampled_oracle"] = sampled_oracle_check(env, counter, bases, offsets, n, L, k)
        check["ok"] = check["ok"] and check["sampled_oracle"]["ok"]
    if check is not None:
        extra["output_check"] = check
        if wl["kind"] == "oligo" and k == 4:   # the launch shape this process measured for itself (kt_oligo_launch_info)
            info = ctx.oligo_launch_info()
            info["ns_per_read"] = {str(a): round(b, 4) for a, b in info["ns_per_read"].items()}
            extra["oligo_launch"] = info
        if not check["ok"]:
            sys.exit("bench.py: %s produced a wrong output: %r" % (name, check))

    if wl["kind"] == "ctr":
        extra["distinct_rank0"] = state["distinct"] if not args.no_export else counter.size_local()
        extra["table_slots_rank0"] = counter.table.capacity()
        if world > 1 or os.environ.get("KT_SHARD_FORCE"):
            rec, km = counter.sharded.route_stats()   # (rank 0's last batch: what went into every owner's region)
            if len(km) and km.sum():
                extra["route"] = {"owners": int(len(km)), "records": int(rec.sum()), "kmers_per_record": round(float(km.sum()) / max(1, int(rec.sum())), 3),
                                  "owner_kmers_max_over_mean": round(float(km.max()) / float(km.mean()), 4)}
        if world == 1 and os.environ.get("KT_SHARD_FORCE"):
            extra["exchanged_bytes_per_rank"] = counter.sharded.exchanged_bytes() // (steps + warmup + ramp)
            extra["exchanged_bytes_note"] = ("one rank routing into KT_SHARD_FORCE owners' regions: the bytes of the regions a "
                                             "rank of that many would have sent")
        if world > 1:
            # did the library's own communicator see `world` ranks, and what did a rank put on the wires per step?
            ci = counter.sharded.comm_info()
            extra["rccl_ranks"] = ci["rccl_ranks"]
            extra["transport"] = ci["transport"]
            extra["exchanged_bytes_per_rank"] = counter.sharded.exchanged_bytes() // (steps + warmup + ramp)
            if not env.share_gpu and ci["rccl_ranks"] != world:
                sys.exit("bench.py: %d ranks were launched but the library's RCCL communicator reports %d (transport %s)"
                         % (world, ci["rccl_ranks"], ci["transport"]))
    res = {
        "value": round(value, 3),
        "unit": "Gbases/s",
        "ms_per_step": round(ms_per_step, 4),
        "ms_median": round(statistics.median(per_step_ms), 4),
        "ms_min": round(min(per_step_ms), 4),
        "ramp_steps": ramp,
        "dtype": wl.get("dtype", "u64 keys / u32 counts"),
        "config": {"workload": wl["desc"] + (", reads sampled from a random %d bp genome (1 %% substitutions)" % genome
                                             if genome else ""),
                   "reads_per_gpu": n, "read_len": L, "k": k, "parallelism": parallelism, "reduced": reduced},
        "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                     # (nothing in this run measures HBM traffic: the figure is read from a committed file)
                     "traffic_source": ("profiles/traffic.json: FETCH_SIZE x 2 + WRITE_SIZE from separate rocprofv3 --pmc "
                                        "passes of this command, committed; not measured in this run") if traffic else None,
                     "traffic_profile": traffic_file, "traffic_profile_head": traffic_head,
                     "kernel": dominant, "kernel_ms": round(kern_ms, 4),
                     "algorithmic_bytes_per_launch": alg_bytes_per_launch},
    }
    if len(first_ms) > 1:
        res["ms_first_20_unramped"] = round(sum(first_ms) / len(first_ms) / launches_per_step, 4)
    pl = extra.get("output_placement")
    if pl and pl.get("ms") and pl["ms"][pl["picked"]] > 0:
        # what a plain allocation (candidate 0 of the placement probes) would have given, by the probes' own ratio
        res["roofline"]["frac_plain_allocation"] = round(achieved / HBM_PEAK_GBS * pl["ms"][pl["picked"]] / pl["ms"][0], 4)
    res.update(extra)
    finish()
    del bases, offsets
    torch.cuda.empty_cache()
    wl["_check_slice"] = check_slice
    return res, wl


def sampled_oracle_check(env, counter, bases, offsets, n, L, k, sample_reads=20000):
    """untimed: the first `sample_reads` reads of rank 0's batch, counted by the CPU oracle, against the shards of all ranks"""
    import numpy as np
    torch, device = env.torch, env.device
    world, rank = env.world, env.rank
    on_cpu = world > 1 and env.share_gpu
    dev = "cpu" if on_cpu else "cuda"
    m = min(n, sample_reads)
    if rank == 0:
        from oracle import kt_oracle as oracle
        hb = bases[:m * L].cpu().numpy()
        ho = offsets[:m + 1].cpu().numpy().astype(np.uint64)
        sk, sc = oracle.count_reads(hb, ho, k, n_parts=4, threads=4)
        cnt = torch.tensor([len(sk)], dtype=torch.int64, device=dev)
    else:
        sk = sc = None
        cnt = torch.zeros(1, dtype=torch.int64, device=dev)
    if world > 1:
        env.dist.broadcast(cnt, src=0)
    nk = int(cnt.item())
    keys = torch.from_numpy(sk.view(np.int64).copy()).to(dev) if rank == 0 else torch.empty(nk, dtype=torch.int64, device=dev)
    if world > 1:
        env.dist.broadcast(keys, src=0)
    dkeys = keys.cuda()
    got = torch.zeros(nk, dtype=torch.int32, device="cuda")
    counter.table.lookup(dkeys, nk, got)
    torch.cuda.synchronize()
    found = (got > 0).to(torch.int32)
    # the keys this rank holds must be its own by the library's host function (a bounded number of them: one ctypes call each)
    mine = dkeys[found.bool()][:2000].cpu().numpy().view(np.uint64)
    stray = sum(1 for x in mine if world > 1 and device.shard_owner_of(int(x), k, world) != rank)
    tot = torch.stack([found.to(torch.int64), got.to(torch.int64)]).to(dev)
    strays = torch.tensor([stray], dtype=torch.int64, device=dev)
    if world > 1:
        env.dist.all_reduce(tot)
        env.dist.all_reduce(strays)
    res = {"reads": m, "keys": nk}
    if rank == 0:
        want = torch.from_numpy(sc.astype(np.int64)).to(dev)
        res["found"] = int((tot[0] == 1).sum().item())
        res["found_on_several_ranks"] = int((tot[0] > 1).sum().item())
        res["count_below_sample"] = int((tot[1] < want).sum().item())
    else:
        res.update(found=nk, found_on_several_ranks=0, count_below_sample=0)
    res["on_a_rank_that_does_not_own_them"] = int(strays.item())
    res["against"] = "CPU oracle (oracle/kt_oracle.c) over the first reads of rank 0's batch"
    res["ok"] = (res["found"] == nk and res["found_on_several_ranks"] == 0 and res["count_below_sample"] == 0
                 and res["on_a_rank_that_does_not_own_them"] == 0)
    return res


def oracle_slice_check(cs):
    """the CHECK_ROWS rows run_workload kept, against the CPU oracle (part of the cpu_baseline leg: the only place
    bench.py touches oracle/): f64 rows bit for bit, f32 rows within north_star's 1e-6"""
    import numpy as np
    from oracle import kt_oracle as oracle
    want = oracle.oligo_batch(cs["bases"], cs["offsets"].astype(np.uint64), cs["k"], True, True, 1.0)
    got = cs["out"]
    if cs["dtype"] == "f64":
        ok = bool(np.array_equal(got.view(np.uint64), want.view(np.uint64)))
        how = "bit-exact"
    else:
        ok = bool(np.abs(got.astype(np.float64) - want).max() <= 1e-6)
        how = "max abs difference <= 1e-6"
    if not ok:
        sys.exit("bench.py: %d-row slice differs from the CPU oracle" % cs["rows"])
    return {"rows": cs["rows"], "against": "CPU oracle (oracle/kt_oracle.c)", "criterion": how, "ok": ok}


def launch_ranks(args):
    """`python bench.py --gpus N` (N > 1) typed as it stands: start N ranks of this same command line through
    torch.distributed.run as a CHILD process - before this process has imported torch or made any HIP call (it never
    does: replacing a process that has initialised the GPU takes the box down, and nothing here needs to) -, pass the
    child's output through and leave with its exit code.  A launch that already comes from torch.distributed.run (the
    driver's own command: WORLD_SIZE is set) does not come through here."""
    import socket
    import subprocess
    port = os.environ.get("MASTER_PORT")
    if not port:
        with socket.socket() as sock:   # a free port of this host
            sock.bind(("127.0.0.1", 0))
            port = str(sock.getsockname()[1])
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", port, str(pathlib.Path(__file__).resolve())] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True)
    for line in proc.stdout:   # rank 0's JSON line (and nothing else: the ranks' diagnostics go to stderr)
        sys.stdout.write(line)
        sys.stdout.flush()
    return proc.wait()


def main():
    args = parse()
    kt_env = kt_environment()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args))
    env = Env(args)
    names = HEADLINE if args.workload == "headline" else (args.workload,)
    results = []
    for name in names:
        res, wl = run_workload(env, name, args, genome=args.genome if WORKLOADS[name]["kind"] in ("ctr", "cov") else 0)
        if (args.workload == "headline" and name == "ctr_k31" and not args.no_genome and not args.genome):
            # the second read distribution for ctr (uniform reads make nearly every 31-mer unique)
            g, _ = run_workload(env, name, args, genome=GENOME_LEN, steps=min(args.steps, 5), warmup=min(args.warmup, 2))
            res["genome_sampled"] = {key: g[key] for key in ("value", "ms_per_step", "ms_median", "ms_min", "roofline",
                                                             "distinct_rank0")}
            res["genome_sampled"]["genome_len"] = GENOME_LEN
            res["genome_sampled"]["steps"] = min(args.steps, 5)
        results.append((name, res, wl))

    if env.rank == 0:
        name, top, wl = results[0]
        line = {"metric": METRIC, "value": top["value"], "unit": "Gbases/s", "n_gpus": env.world, "steps": args.steps,
                "warmup": args.warmup, "ms_per_step": top["ms_per_step"], "higher_is_better": True, "scaling": "weak",
                "vs_baseline": None, "dtype": top["dtype"], "data": "synthetic"}
        line.update({key: v for key, v in top.items() if key not in line})
        line["kt_env"] = kt_env
        if env.world == 1 and not args.no_cpu:
            line["cpu_baseline"] = cpu_baseline_for(wl, args, args.genome)
            if wl.get("_check_slice"):
                line["output_check"]["oracle_slice"] = oracle_slice_check(wl["_check_slice"])
        for name, res, wl in results[1:]:
            obj = dict(res)
            obj.update({"n_gpus": env.world, "steps": args.steps, "warmup": args.warmup, "scaling": "weak"})
            if env.world == 1 and not args.no_cpu:
                if name in ("ctr_k31",):   # the headline's second half has its own CPU leg; the extra configs do not
                    obj["cpu_baseline"] = cpu_baseline_for(wl, args, 0)
                if wl.get("_check_slice"):
                    obj["output_check"]["oracle_slice"] = oracle_slice_check(wl["_check_slice"])
            line[name] = obj
        print(json.dumps(line), flush=True)
    env.close()


if __name__ == "__main__":
    main()
