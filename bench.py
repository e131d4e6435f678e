#!/usr/bin/env python
"""bench.py - walker-lnprob evaluations per second of the HIP path (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W          (N > 1: starts its N ranks itself)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Workload (config.workload): the configuration the metric is quoted on - 10^6 synthetic sources,
256 walkers, single-Schechter model with free completeness (LumFuncMCMC.lnprob, the drivers'
default), fp64.  A step is one ensemble step: two half-ensemble calls of 128 theta rows each
(what emcee's stretch move issues), i.e. 256 walker-lnprob evaluations over the whole catalogue.
Catalogue, grids and theta blocks are resident in HBM before the timed region.  One JSON line on
stdout (rank 0).

Several GPUs (one process per GPU, torch.distributed, backend nccl = RCCL over xGMI):
  --scaling weak   (default) every rank brings --walkers walkers: N x 256 walkers in all, each half-step is
                   sharded by walker and ends with the all-gather of the per-walker lnprob.
  --scaling strong the ensemble is fixed (--walkers in all: 256 for the metric, 1024 for BASELINE config 4,
                   512 zevol for config 5) and split over the ranks,
      --shard walkers   by walker: rank r evaluates rows r*B/N .. of every block, all-gather;
      --shard sources   by source: rank r holds 1/N of every field's sources and of the grid chunks and
                        evaluates ALL rows, all-reduce(SUM) of B doubles;
      --shard auto      sources when a rank's slice of a block would be under four 16-walker tiles.
With N > 1 and weak scaling the line also carries "strong_scaling": the metric's fixed 256-walker
ensemble timed in the same run, both ways of sharding.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP64_VALU_PEAK_TFLOPS = 78.6      # MI355X vector fp64 (spec; half of the 157.3 fp32 figure of MI355X_MICROARCH.md)
HBM_PEAK_GBS = 8000.0             # MI355X_MICROARCH.md: 8 TB/s spec
# EXECUTED fp64 flops and issue cycles per (walker, source) term / per (walker, node, field) term, by FORM of the term,
# come from profiles/isa_counts.json, which lumfuncmcmc_amd.build writes from the compiler's assembly of the same
# sources (profiles/isa_mix.py: FMA = 2 flops, any other fp64 VALU instruction = 1); HOW MANY items took which form
# in the timed workload comes from the kernel's own census (lf_form_counts).  (The survey's nominal weights - exp/log
# = 40 flops - would put the same run above peak; DESIGN.md section 4 explains why that figure is not used.)
SETTLE_STEPS = 20                 # untimed steps after the W warm-up steps, see timed()
SETTLE_SECONDS = 0.25             # ... and at least this long under load
SOURCE_FORMS = ("general", "general_noexp", "table", "table_noexp", "cell")      # ("cell": per (walker, cell), not per source)
NODE_FORMS = ("node_general", "node_bright")
# the other variants' grid parts (one or two exponentials per node) and the careful path are left out of `achieved`
# bytes the per-source loop streams per source and launch: logf_i and U_i = 10^(logf_i + 17) (free: the
# Schechter part is closed-form per walker, so lum_i is not read); lum, z, z^2 (zevol); nothing (fixcomp).
# SURVEY.md section 8d also counts 16 B for the free variant.
BYTES_PER_SOURCE = {"free": 16, "fixcomp": 0, "zevol": 24}


def isa_counts(launch, variant):
    """Per-form {flops_per_item, cycles_per_item} of the lf_main instantiation that ran (profiles/isa_counts.json)."""
    doc = json.load(open(os.path.join(ROOT, "profiles", "isa_counts.json")))
    vi = {"free": 0, "fixcomp": 1, "zevol": 2}[variant]
    if launch.get("kind") == 2:                                       # (capi: the fused form reports kind 2 and "fused")
        key = "lf_free<%d>" % launch["st"]
    elif launch.get("kind") == 4:                                     # the persistent kernel of the other two variants
        key = "lf_pers<%d>" % vi
    else:
        key = "lf_main<%d,%d,%d,%d,%s>" % (vi, launch["st"], launch["tw"], launch["twb"], "true" if launch["compressed"] else "false")
    forms = doc["kernels"].get(key, {})
    if launch["compressed"] or not forms:
        # the compressed-catalogue instantiations carry no markers of their own: price with the direct kernel's forms
        forms = doc["kernels"].get("lf_main<%d,8,16,16,false>" % vi, {})
    return key, forms


def build_model(variant, nsrc, walkers, device):
    from lumfuncmcmc_amd import synth
    from lumfuncmcmc_amd.model import LumFuncMCMC, LumFuncMCMCz
    cat = synth.catalogue(nsrc, seed=20241016, zslices=8 if variant == "zevol" else 0)
    fi = cat["field_ind"]
    kw = dict(lum=synth.split_fields(cat["lum"], fi), lum_e=synth.split_fields(cat["lum_e"], fi),
              Flim=list(synth.FLIM), alpha=synth.ALPHA_C, Omega_0=list(synth.OMEGA_0), sch_al=synth.SCH_AL,
              sch_al_lims=synth.SCH_AL_LIMS, Lstar=synth.LSTAR, Lstar_lims=synth.LSTAR_LIMS,
              phistar=synth.PHISTAR, phistar_lims=synth.PHISTAR_LIMS, Lc=synth.LC, Lh=synth.LH,
              nwalkers=walkers, nsteps=1, min_comp_frac=synth.MIN_COMP_FRAC, field_ind=fi, device=device)
    zs = synth.split_fields(cat["z"], fi)
    if variant == "zevol":
        return LumFuncMCMCz(zs, **kw)
    return LumFuncMCMC(zs, fix_comp=(variant == "fixcomp"), Flim_lims=synth.FLIM_LIMS,
                       alpha_lims=synth.ALPHA_LIMS, **kw)


def cpu_baseline(model, variant, theta, budget_s=20.0):
    """The NumPy port (oracle/lf_oracle.py) on this host, one thread, one theta row per call -
    the reference's execution model under emcee.  Bounded sample of the same workload."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import lf_oracle as O
    inp = model.kernel_inputs()
    inp["lims"] = {k: list(v) for k, v in inp["lims"].items()}
    n, t0, vals = 0, time.perf_counter(), []
    while n < 2 or time.perf_counter() - t0 < budget_s:     # cycle over the block until the budget is spent
        v = O.lnprob(inp, theta[n % len(theta)])
        if n < len(theta):
            vals.append(v)
        n += 1
    dt = time.perf_counter() - t0
    return {"value": n / dt, "unit": "walker-lnprob evals/s", "cores": 1, "kind": "port",
            "sample": "%d evaluations cycling over the theta rows of the timed workload, %.1f s, single-thread NumPy scalar loop "
                      "(DL(z_i) hoisted out of the call, which the reference does not do)" % (n, dt)}, np.array(vals)


def cpu_baseline_allcores(model, theta, nthreads):
    """The C restatement (oracle/lf_oracle.c, OpenMP over theta rows) on all the host cores this process
    may use: the second CPU figure SURVEY.md section 8d asks for.  Bounded: 4 rows per thread."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from lf_oracle_c import COracle
    inp = model.kernel_inputs()
    inp["lims"] = {k: list(v) for k, v in inp["lims"].items()}
    co = COracle(inp)
    rows = np.resize(theta, (4 * nthreads, theta.shape[1]))
    co.lnprob_batch(rows[:nthreads], nthreads=nthreads)            # warm
    t0 = time.perf_counter()
    co.lnprob_batch(rows, nthreads=nthreads)
    dt = time.perf_counter() - t0
    return {"value": len(rows) / dt, "unit": "walker-lnprob evals/s", "cores": nthreads, "kind": "port",
            "sample": "%d theta rows of the timed workload over %d OpenMP threads, %.1f s, plain-C scalar loop" % (len(rows), nthreads, dt)}


def self_launch(n):
    """`python bench.py --gpus N` with no WORLD_SIZE: start the N ranks as children (torch.distributed.run, one
    process per GPU) BEFORE this process imports torch or touches the GPU, relay rank 0's JSON line, and exit
    non-zero if any rank failed.  Nothing is exec'ed."""
    import socket
    import subprocess
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC (RCCL across processes), see the box's notes
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE)
    lines = [l for l in r.stdout.decode(errors="replace").splitlines() if l.startswith("{")]
    if r.returncode != 0 or not lines:
        sys.stderr.write("bench.py: the %d-rank run failed (exit code %d, %d JSON lines)\n" % (n, r.returncode, len(lines)))
        sys.exit(r.returncode or 1)
    sys.stdout.write(lines[-1] + "\n")
    sys.stdout.flush()
    sys.exit(0)


def compressed_leg(ctx, step, fence, direct_out, steps, W):
    import gc
    import torch
    t0 = time.perf_counter()
    ctx.set_option("compress", 1)
    build_s = time.perf_counter() - t0
    for i in range(3):
        out = step(i)
    fence()
    gc.collect()
    gc.disable()                                    # as in main(): no collector pause inside a timed window
    t0 = time.perf_counter()
    for i in range(steps):
        out = step(i)
    fence()
    dt = time.perf_counter() - t0
    gc.enable()
    rel = float(torch.max(torch.abs(out[0] - direct_out[0]) / torch.abs(direct_out[0])).item())
    res = {"value": W * steps / dt, "unit": "walker-lnprob evals/s", "ms_per_step": dt / steps * 1e3,
           "build_s": build_s, "max_rel_diff_vs_direct": rel,
           "note": "opt-in option, off by default; piece A over weighted pseudo-sources and (free variant, separable grid) "
                   "piece B over shared flux nodes; bins validated at sampled walkers of the prior box (not a proven bound) "
                   "and self-checked against the direct path at build time"}
    # the same catalogue with 2048 walkers (cost per evaluation no longer depends on N: more walkers fill the GPU)
    try:
        from lumfuncmcmc_amd import synth
        from lumfuncmcmc_amd.sampler import DeviceEnsembleSampler
        variant = ctx.variant
        if variant in ("free", "zevol"):
            W2, n2 = 2048, max(20, steps)
            ds = DeviceEnsembleSampler(ctx, W2, seed=1, capacity=n2 + 60)
            # burn in first: proposals from the broad start often land where the per-source underflow checks are needed,
            # and those walkers are summed over the real catalogue (rescue workgroups) - a transient of the first steps
            ds.run_mcmc(synth.walkers(variant, W2, seed=4), 60)
            t0 = time.perf_counter()
            ds.run_mcmc(None, n2)
            t2 = time.perf_counter() - t0
            ds.close()
            res["device_sampler_2048_walkers"] = {"value": W2 * n2 / t2, "unit": "walker-lnprob evals/s", "ms_per_step": t2 / n2 * 1e3}
    except Exception as e:                     # an extra figure: never at the cost of the bench line
        res["device_sampler_2048_walkers"] = {"error": str(e)}
    ctx.set_option("compress", 0)
    return res


def resolve_sharding(args, world):
    """(scaling, shard) actually run.  Weak scaling is always walker-sharded (more walkers, same catalogue)."""
    if world == 1 and not args.force_collective:
        return args.scaling, "walkers"
    if args.scaling == "weak":
        return "weak", "walkers"
    shard = args.shard
    if shard == "auto":
        rows_per_rank = -(-(args.walkers // 2) // world)
        shard = "sources" if rows_per_rank < 64 else "walkers"          # under four 16-walker tiles per call
    return "strong", shard


class Leg(object):
    """One timed configuration: (total walkers, sharding) -> evaluator over this rank's context."""

    def __init__(self, args, model, dev, local, world, rank, Wtot, shard, nblk=8, seed=1):
        import torch
        from lumfuncmcmc_amd import synth
        from lumfuncmcmc_amd.dist import ShardedLnProb, SourceShardedLnProb, shard_sources, slice_bounds
        self.args, self.world, self.rank, self.shard, self.Wtot = args, world, rank, shard, Wtot
        self.defer = False
        self.half = Wtot // 2
        self.nblk = nblk
        ki = model.kernel_inputs()
        if shard == "sources" and world > 1:
            self.ev = SourceShardedLnProb(ki, local)
            self.ctx = self.ev.ctx
            self.own_ctx = True
            self.ki = shard_sources(ki, rank, world)
            self.rows_local, self.row_lo = self.half, 0
            self.grid_share = (rank, world)
        else:
            t0 = time.perf_counter()
            self.ctx = model.context()          # lf_create: the flux sort, the cells, the proven grid bins, the tables' upload
            self.create_s = time.perf_counter() - t0
            self.own_ctx = False
            self.ev = ShardedLnProb(self.ctx.lnprob_torch, self.ctx.ndim, dev, force_collective=args.force_collective)
            self.defer = (world > 1 or args.force_collective) and args.pipeline_collective
            self.ki = ki
            bounds, _ = slice_bounds(self.half, world)
            self.row_lo, hi = bounds[rank]
            self.rows_local = hi - self.row_lo
            self.grid_share = (0, 1)
        ctx = self.ctx
        if args.geometry >= 0:
            ctx.set_option("geometry", args.geometry)
        if args.walker_tile:
            ctx.set_option("walker_tile", args.walker_tile)
        if args.taper:
            ctx.set_option("taper", 1)
        if args.compress:
            ctx.set_option("compress", 1)
        if args.no_specialise:
            ctx.set_option("specialise", 0)
        if args.no_tables:
            ctx.set_option("tables", 0)
        if args.no_cells:
            ctx.set_option("cells", 0)
        if args.no_fuse:
            ctx.set_option("fuse", 0)
        if args.no_grid_shortcut:
            ctx.set_option("grid_shortcut", 0)
        self.ndim = ctx.ndim
        # global half-ensemble blocks, identical on every rank; 4 distinct steps' worth, cycled
        self.theta_all = synth.walkers(args.variant, self.half * nblk, seed=seed).reshape(nblk, self.half, self.ndim)
        self.blocks = [torch.from_numpy(self.theta_all[i]).to(dev) for i in range(nblk)]

    def step(self, i):
        if self.defer:
            # the blocks of the plain loop are independent: the all-gather of one overlaps the kernel of the next
            # (dist.ShardedLnProb: defer; complete at the fence, which flushes)
            a = self.ev.evaluate_tensor(self.blocks[(2 * i) % self.nblk], defer=True)
            b = self.ev.evaluate_tensor(self.blocks[(2 * i + 1) % self.nblk], defer=True)
            return a, b
        a = self.ev.evaluate_tensor(self.blocks[(2 * i) % self.nblk])
        b = self.ev.evaluate_tensor(self.blocks[(2 * i + 1) % self.nblk])
        return a, b

    def flush(self):
        if self.defer:
            self.ev.flush()

    def local_rows(self, used):
        """theta rows of the blocks `used` that THIS rank's kernels evaluate."""
        return [self.theta_all[j, self.row_lo:self.row_lo + self.rows_local] for j in used]

    def close(self):
        if self.own_ctx:
            self.ev.close()


def fence_light(leg):
    """Keeps the host from running far ahead of the device during the settle phase (local, no collective)."""
    import torch
    torch.cuda.current_stream().synchronize()


def timed(leg, fence, warmup, steps, profile_level, agree=None, profile_every=1, profile_span=1):
    """W warm-up + settle steps, then `steps` timed steps between two fences.  Returns (seconds, kernel times, last out)."""
    import gc
    for i in range(warmup):
        out = leg.step(i)
    # no interpreter pauses inside the timed window: with torch imported a full collection walks ~10^6 objects
    # (40-80 ms, and whether one falls into the window depends on how many objects the flags allocated before)
    gc.collect()
    gc.disable()
    # settle: more untimed steps after whatever warm-up was asked for, until the device has been under this load for
    # SETTLE_SECONDS.  The power management of the GPU takes tens of milliseconds of sustained work to reach the
    # clocks it then holds (tools/time_parts.py: the same launch 127 us in the first 10 ms of load, 112 us after 30 ms);
    # an MCMC run lasts minutes, so the steady state is the one to quote.  Nothing idles between here and the timed
    # steps: the fence only waits for the queue to drain.
    # (Every rank runs the SAME number of steps - they hold collectives: the count comes from a timed first batch and is
    # agreed by max over the ranks.)
    t_settle = time.perf_counter()
    for i in range(SETTLE_STEPS):
        out = leg.step(i)
    leg.flush()
    fence()
    per_step = (time.perf_counter() - t_settle) / SETTLE_STEPS
    more = int(min(SETTLE_SECONDS / max(per_step, 1e-6), 20000))
    if agree is not None:
        more = int(agree(float(more)))
    for i in range(more):
        out = leg.step(i)
        if i % 64 == 63:
            fence_light(leg)
    leg.flush()
    fence()
    leg.ctx.kernel_times()                      # clear
    # The event pairs that time the kernel are barrier packets: a bracketed launch no longer overlaps the tail of the one
    # before it (tools/call_period.py: 39.7 us per evaluation without events, 47.3 us with a pair around every lf_main).
    # So only every `profile_every`-th evaluation of the timed steps is bracketed - still live, still in the timed
    # region, on the launch stream; "launches" in the roofline object is the number that were.
    # ... and where an evaluation is ONE launch, a pair goes around a run of `profile_span` consecutive evaluations: the run's
    # average is a launch's duration as the stream sees it (a pair around a single launch adds its dispatch: 16.6 us
    # where rocprofv3 and the period of back-to-back calls say 13.6)
    leg.ctx.set_option("profile_every", max(int(profile_every), 1))
    leg.ctx.set_option("profile_span", max(int(profile_span), 1))
    leg.ctx.set_profiling(profile_level)
    t0 = time.perf_counter()
    dbg = []
    for i in range(steps):
        out = leg.step(i)
        dbg.append(time.perf_counter() - t0)
    leg.flush()
    fence()
    dt = time.perf_counter() - t0
    gc.enable()
    if os.environ.get("LF_BENCH_DEBUG"):
        print("debug: host time after each step (ms):", " ".join("%.2f" % (x * 1e3) for x in dbg[:8]), "fence done %.2f" % (dt * 1e3), file=sys.stderr)
    leg.ctx.set_profiling(0)
    kt = leg.ctx.kernel_times()
    leg.ctx.set_option("profile_every", 1)
    leg.ctx.set_option("profile_span", 1)
    return dt, kt, out


def census(leg, used):
    """Which form of the term ran how often, per lf_main launch of THIS rank, over the blocks of the timed workload:
    one extra pass with the kernel's census switched on (outside the timed region; it costs an atomic per
    (walker, chunk))."""
    import torch
    ctx = leg.ctx
    ctx.set_option("count_forms", 1)
    for j in used:                                  # this rank's own launches only: no collective (rank 0 runs this alone)
        rows = leg.blocks[j][leg.row_lo:leg.row_lo + leg.rows_local]
        if rows.shape[0]:
            ctx.lnprob_torch(rows.contiguous())
    torch.cuda.synchronize()
    counts = ctx.form_counts()
    ctx.set_option("count_forms", 0)
    return {k: v / float(len(used)) for k, v in counts.items()}


def grid_rows_per_bin(model):
    """Mean number of luminosity rows that cross a flux bin of piece B (free variant, lf_gridbound.h): the length of the dot
    product a bin's lane makes with the rows' Schechter values.  Recomputed on the host from the model's own grid."""
    from lumfuncmcmc_amd import capi
    ki = model.kernel_inputs()
    logL, zarr = np.asarray(ki["logL"]), np.asarray(ki["zarr"])
    S = logL.shape[0]
    if not np.all(logL == logL[:, :1]):
        return None
    L = logL[:, 0]
    wL = np.gradient(L) * 1.0
    wL[0], wL[-1] = 0.5 * (L[1] - L[0]), 0.5 * (L[-1] - L[-2])
    wz = np.gradient(zarr)
    wz[0], wz[-1] = 0.5 * (zarr[1] - zarr[0]), 0.5 * (zarr[-1] - zarr[-2])
    Dk = np.log10(4.0 * np.pi * (capi.MPC_CM * np.asarray(ki["DL_zarr"])) ** 2)
    fcmin = float(ki.get("fcmin", 0.1))
    fr = abs((2 * fcmin - 1) ** 2 / (1 - (2 * fcmin - 1) ** 2))
    try:
        g = capi.grid_bins([fr, ki["lims"]["alpha"][0], ki["lims"]["alpha"][1], ki["lims"]["Flim"][0], ki["lims"]["Flim"][1]],
                           L, wL, wz * np.asarray(ki["volume_part"]), Dk)
    except RuntimeError:
        return None
    return float(np.mean(g["rows"][:, 1])), int(g["rows"].shape[0])


def roofline_of(args, leg, model, kt, dt):
    """fp64-VALU roofline of the dominant kernel on THIS rank: EXECUTED flops of its share of the work / its average launch
    duration (HIP events on the launch stream).  Executed flops = how many items took which form of the arithmetic (the
    kernel's census, or - the persistent kernel of the other two variants - the chunk counts of its launch) x the flops of
    that form in the compiler's assembly of the same sources (profiles/isa_counts.json)."""
    variant = args.variant
    k = kt["main"]
    launches = max(k["launches"], 1)
    avg_ms = k["ms"] / launches if k["launches"] else dt / args.steps / 2 * 1e3   # no events: the whole call
    nsrc, rows = leg.ctx.N, leg.rows_local
    terms = float(nsrc) * rows                                        # (walker, source) terms per launch
    used = sorted({(2 * i) % leg.nblk for i in range(args.steps)} | {(2 * i + 1) % leg.nblk for i in range(args.steps)})
    launch = leg.ctx.last_launch()
    kernel, forms = isa_counts(launch, variant)
    zero = {"flops_per_item": 0.0, "cycles_per_item": 0.0}
    per = lambda f: forms.get(f, zero)
    cnt, extra = {}, {}
    if launch["kind"] == 4:
        # lf_pers: every walker inside the prior is summed over the cells in redshift (z-evolving) and integrates every node
        # of the grid (64 per chunk, pads weigh 0); piece A of the fixed-completeness variant is a closed form
        cnt = {"pzcell": 64.0 * launch["chunks_a"] * rows, "pznode": 64.0 * launch["chunks_b"] * rows}
        src_flops, src_cycles = cnt["pzcell"] * per("pzcell")["flops_per_item"], cnt["pzcell"] * per("pzcell")["cycles_per_item"]
        grid_flops, grid_cycles = cnt["pznode"] * per("pznode")["flops_per_item"], cnt["pznode"] * per("pznode")["cycles_per_item"]
    elif variant == "zevol":
        cnt = census(leg, used)
        # lf_main's census: "cell" = (walker, cell in redshift) pairs, "table" = terms of the local form (one exponential per
        # lane of z-neighbours), "general" = per-source exponentials, "node_general" = grid nodes
        name = {"cell": "zcell", "table": "zevol", "general": "zevol_direct", "node_general": "znode"}
        src_flops = sum(cnt.get(f, 0.0) * per(name[f])["flops_per_item"] for f in ("cell", "table", "general"))
        src_cycles = sum(cnt.get(f, 0.0) * per(name[f])["cycles_per_item"] for f in ("cell", "table", "general"))
        grid_flops = cnt.get("node_general", 0.0) * per("znode")["flops_per_item"]
        grid_cycles = cnt.get("node_general", 0.0) * per("znode")["cycles_per_item"]
    elif variant == "free":
        cnt = census(leg, used)
        src_flops = sum(cnt.get(f, 0.0) * per(f)["flops_per_item"] for f in SOURCE_FORMS)
        src_cycles = sum(cnt.get(f, 0.0) * per(f)["cycles_per_item"] for f in SOURCE_FORMS)
        grid_flops = sum(cnt.get(f, 0.0) * per(f)["flops_per_item"] for f in NODE_FORMS)
        grid_cycles = sum(cnt.get(f, 0.0) * per(f)["cycles_per_item"] for f in NODE_FORMS)
        rb = grid_rows_per_bin(model) if launch["kind"] == 2 and launch["chunks_b"] <= 64 else None
        if rb is not None:
            # piece B over flux bins: per (walker, bin, lane) besides the node's completeness sum (census: node_*) one
            # Schechter value (an exponential: "qT") and a dot product over the rows that cross the bin (2 flops per row)
            lanes = (cnt.get("node_general", 0.0) + cnt.get("node_bright", 0.0)) / max(leg.ctx.nf, 1)
            extra = {"bin_lanes": lanes, "rows_per_bin": rb[0], "bins": rb[1]}
            grid_flops += lanes * (per("qT")["flops_per_item"] + 2.0 * rb[0])
            grid_cycles += lanes * (per("qT")["cycles_per_item"] + 4.0 * rb[0])
    else:
        # fixed completeness in lf_main: piece A is a closed form, the grid costs one exponential per node
        nodes = float(launch["chunks_b"]) * 256.0 * rows
        cnt = {"fixnode": nodes}
        src_flops = src_cycles = 0.0
        grid_flops, grid_cycles = nodes * per("fixnode")["flops_per_item"], nodes * per("fixnode")["cycles_per_item"]
    alg_flops = src_flops + grid_flops
    alg_bytes = nsrc * BYTES_PER_SOURCE[variant] + rows * 8 * (leg.ndim + 1)
    on_cells = cnt.get("cell", 0.0) > 0.0 and cnt.get("table", 0.0) + cnt.get("general", 0.0) == 0.0
    if on_cells or launch["kind"] == 4:
        # every walker was summed over the catalogue's cells (or, fixed completeness, in closed form): what a launch streams
        # is the cells' records (80 B free, 64 B z-evolving) and the grid's nodes / bins' nodes, once
        if launch["kind"] == 4:
            alg_bytes = 64.0 * launch["chunks_a"] * 64 + 64.0 * launch["chunks_b"] * 32 + rows * 8 * (leg.ndim + 1)
        else:
            nodes = (cnt.get("node_general", 0.0) + cnt.get("node_bright", 0.0)) / max(rows, 1) / (leg.ctx.nf if variant == "free" else 1)
            alg_bytes = cnt["cell"] / max(rows, 1) * (80 if variant == "free" else 64) + nodes * (64 if variant == "free" else 40) \
                + rows * 8 * (leg.ndim + 1)
    traffic = rocprof_ms = None
    tf = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    ab = [n for n in AB_FLAGS if getattr(args, n)]
    if os.path.exists(tf) and leg.world == 1 and not ab:
        try:
            ent = json.load(open(tf)).get("%s_n%d_b%d" % (variant, nsrc, rows), {})
            traffic = ent.get("hbm_bytes_per_launch")
            # (the committed rocprofv3 --kernel-trace --stats run of this same command: the kernel between its own begin
            # and end; the event pair here also spans the dispatch of the launch)
            rocprof_ms = ent.get("rocprofv3_avg_launch_ns") and ent["rocprofv3_avg_launch_ns"] * 1e-6
        except Exception:
            traffic = rocprof_ms = None
    ach_tf = alg_flops / (avg_ms * 1e-3) / 1e12
    ach_gb = alg_bytes / (avg_ms * 1e-3) / 1e9
    nominal = None
    if variant == "free":
        # SURVEY.md section 8d's NOMINAL count (every (walker, source) term at 226 flops, every node-field term at 228 + ...):
        # what the reference's formulas would cost evaluated term by term - the kernel does not do that work (closed-form
        # Schechter sum, cells, flux bins), so this figure is not a rate of anything executed
        S = leg.ctx.S
        nominal = rows * (nsrc * 226.0 + S * S * (46.0 + leg.ctx.nf * 182.0)) / (avg_ms * 1e-3) / 1e12
    note = None
    if on_cells or launch["kind"] == 4:
        note = ("frac prices EXECUTED fp64 flops (census / chunk counts x the ISA's flops per item) against the vector-fp64 peak.  "
                "The launch is a latency chain, not a stream of arithmetic: kernel arguments and theta from a cold cache, the "
                "walkers' preparation, a few chunks of cells and grid per wave, the write-through of the partial sums, a counter, "
                "the last workgroup's sums (tools/stamps_fused.py, profiles/*stamps*): frac says how little of the launch is "
                "arithmetic, not how well the arithmetic runs." +
                (("  SURVEY section 8d's NOMINAL count (226 flops per (walker, source) term, 228 per node-field term) would read "
                  "%.0f TFLOP/s = %.1f x the peak here: the kernel does not do that work - closed-form Schechter sum, cells (a "
                  "run of flux-neighbouring sources summed from 10 numbers), piece B over 64-node flux bins with a proven bound - "
                  "and the parity tests against the reference's own lnprob are the proof that the result is the same.  The "
                  "per-source kernel that does evaluate every term is in per_source_kernel." % (nominal, nominal / FP64_VALU_PEAK_TFLOPS))
                 if nominal else ""))
    return {"bound": "valu_fp64", "kernel": kernel, "achieved": ach_tf,
            "peak": FP64_VALU_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach_tf / FP64_VALU_PEAK_TFLOPS,
            "traffic": traffic, "avg_launch_ms": avg_ms, "launches": launches, "rocprofv3_avg_launch_ms": rocprof_ms,
            "traffic_and_rocprofv3_from": "profiles/hbm_traffic.json (the committed rocprofv3 passes of this command; not measured by this run)"
            if traffic is not None else None,
            "launches_timed": ("HIP events on the launch stream: one pair around each run of %d consecutive evaluations (one launch each) starting at "
                               "every %d-th of the timed region's %d; avg_launch_ms is the runs' average per launch"
                               % (args.profile_span, max(args.profile_every, 1), 2 * args.steps)) if args.profile_span > 1 and launch.get("fused")
            else "every %d-th of the timed region's %d, HIP events on the launch stream" % (max(args.profile_every, 1), 2 * args.steps),
            "measured_on": "rank 0",
            "launch": launch, "terms_per_launch": terms,
            "flops_per_launch": {"source_terms": src_flops, "grid_integral": grid_flops},
            "items_per_launch_by_form": cnt, "flux_bins": extra or None,
            "flops_per_item_by_form": {f: forms[f]["flops_per_item"] for f in forms},
            "cycles_per_item_by_form": {f: forms[f]["cycles_per_item"] for f in forms},
            "nominal_survey_8d_tflops": nominal,
            "terms_per_s": terms / (avg_ms * 1e-3),
            # fraction of the 1024 SIMDs' issue cycles (at the 2.4 GHz spec clock) the counted forms need
            "valu_issue_frac_at_2p4GHz": (src_cycles + grid_cycles) / 64.0 / (avg_ms * 1e-3) / (1024 * 2.4e9),
            "hbm": {"bound": "hbm", "achieved": ach_gb, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": ach_gb / HBM_PEAK_GBS, "algorithmic_bytes_per_launch": alg_bytes},
            "kernel_ms": {n: v["ms"] / max(v["launches"], 1) for n, v in kt.items() if n != "unused"},
            "note": note}


AB_FLAGS = ("no_cells", "no_fuse", "no_tables", "no_specialise", "no_grid_shortcut", "compress", "taper")
# BASELINE.json's configurations as presets: variant, sources, walkers in all, scaling ("config 1" is the reference's CPU case:
# its shape runs here on the GPU)
CONFIGS = {1: ("free", 1000, 32, "weak"), 2: ("free", 100000, 256, "weak"), 3: ("free", 1000000, 512, "weak"),
           4: ("free", 1000000, 1024, "strong"), 5: ("zevol", 800000, 512, "strong")}


def host_cost(leg, burst=64, reps=40):
    """Host time of one evaluation call (Python -> ctypes -> hipLaunchKernel), apart from the queue's back-pressure: bursts of
    `burst` calls on an idle stream; and the same evaluations enqueued by ONE call of the C loop (lf_lnprob_batch_device_n)."""
    import gc
    import torch
    ctx = leg.ctx
    th = leg.blocks[0][leg.row_lo:leg.row_lo + leg.rows_local].contiguous()
    out = torch.empty(th.shape[0], dtype=torch.float64, device=th.device)
    ring = th.unsqueeze(0).repeat(burst, 1, 1).contiguous()
    out_n = torch.empty((burst, th.shape[0]), dtype=torch.float64, device=th.device)
    res = {}
    gc.collect()
    gc.disable()
    for name, call, per in (("python_per_call", lambda: [ctx.lnprob_torch(th, out=out) for _ in range(burst)], burst),
                            ("c_loop_per_evaluation", lambda: ctx.lnprob_torch_n(ring, out=out_n), burst)):
        host, total = [], []
        for _ in range(reps):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            call()
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            host.append(1e6 * (t1 - t0) / per)
            total.append(1e6 * (t2 - t0) / per)
        res[name] = {"host_us": float(np.median(host[3:])), "enqueue_plus_device_us": float(np.median(total[3:]))}
    gc.enable()
    res["burst"] = burst
    res["note"] = ("host_us: median over bursts of %d evaluations enqueued on an idle stream (the queue never fills: no "
                   "back-pressure in the figure); enqueue_plus_device_us: the same bursts up to the device's completion" % burst)
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--variant", default="free", choices=["free", "fixcomp", "zevol"])
    ap.add_argument("--nsrc", type=int, default=1000000)
    ap.add_argument("--walkers", type=int, default=256, help="walkers per GPU (weak scaling) or in all (strong scaling)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: --walkers per GPU; strong: --walkers in all, split over the GPUs")
    ap.add_argument("--shard", default="auto", choices=["walkers", "sources", "auto"],
                    help="strong scaling: split every block by walker (all-gather) or the catalogue by source (all-reduce)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget", type=float, default=20.0)
    ap.add_argument("--geometry", type=int, default=-1)
    ap.add_argument("--walker-tile", type=int, default=0)
    ap.add_argument("--no-taper", action="store_true", help="(default) single pass over the catalogue")
    ap.add_argument("--taper", action="store_true", help="quarter-size tail tiles, see the taper option")
    ap.add_argument("--no-extras", action="store_true", help="only the timed loop: no device-sampler / compressed-catalogue / strong-scaling legs (profiling runs)")
    ap.add_argument("--no-specialise", action="store_true", help="A/B: without the chunk-level term specialisation")
    ap.add_argument("--no-fuse", action="store_true", help="A/B: three launches per evaluation (lf_prepare, lf_free, lf_finalize) instead of one")
    ap.add_argument("--no-cells", action="store_true", help="A/B: every walker summed over the sources, not over the catalogue's cells")
    ap.add_argument("--no-tables", action="store_true", help="A/B: the general form of the free term only (no g/h tables)")
    ap.add_argument("--no-grid-shortcut", action="store_true",
                    help="A/B: piece B of the free variant over the S^2 lattice points instead of the proven flux bins")
    ap.add_argument("--config", type=int, default=0, choices=[0, 1, 2, 3, 4, 5],
                    help="BASELINE.json's configuration 1-5 as a preset of --variant / --nsrc / --walkers / --scaling (0: the metric's workload)")
    ap.add_argument("--force-collective", action="store_true",
                    help="one-GPU rehearsal of the multi-GPU path: initialise the process group and run the all-gather with one rank")
    ap.add_argument("--compress", action="store_true", help="time the compressed-catalogue option instead of the direct kernel (not the headline)")
    ap.add_argument("--pipeline-collective", action="store_true",
                    help="several GPUs, walker-sharded: do not wait for a block's all-gather before the next block's kernel (the plain "
                         "loop's blocks are independent).  Off by default: measured with a one-rank RCCL group torch's asynchronous "
                         "collectives cost more than they hide (30.4 against 28.4 us per 128-row call, 138 against 24.5 at 64 rows: "
                         "profiles/r03_collective_rehearsal.txt)")
    ap.add_argument("--default-stream", action="store_true", help="launch on the legacy default stream instead of a side stream")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to rehearse the plumbing)")
    ap.add_argument("--share-gpu", action="store_true", help="rehearsal on a 1-GPU box: every rank uses device 0")
    ap.add_argument("--profile-level", type=int, default=1, help="HIP events: 0 none, 1 around lf_main, 2 every launch")
    ap.add_argument("--profile-span", type=int, default=20,
                    help="one-launch evaluations: one event pair around this many consecutive evaluations (their average); 1 = a pair per launch")
    ap.add_argument("--profile-every", type=int, default=0,
                    help="bracket only every n-th evaluation of the timed steps with events (each pair is two barrier packets and stalls "
                         "the stream 8-12 us - most of an evaluation by now); 0 = as many as give two bracketed launches in the timed steps")
    args = ap.parse_args()
    if args.config:
        args.variant, args.nsrc, wtot, args.scaling = CONFIGS[args.config]
        args.walkers = wtot if args.scaling == "strong" else wtot      # (weak presets are 1-GPU shapes: walkers per GPU = in all)
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args.gpus)                  # never returns; this process has not touched torch or the GPU
    # stdout carries exactly ONE line (the JSON): libraries that print there (RCCL's version banner at
    # communicator init does) are sent to stderr from here on
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    if args.share_gpu:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    multi = world > 1 or args.force_collective
    if multi:
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29517")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)

    scaling, shard = resolve_sharding(args, world)
    W = args.walkers
    Wtot = W * world if scaling == "weak" else W
    if Wtot % 2 or Wtot < 2:
        raise SystemExit("the ensemble needs an even number of walkers")
    t_setup = time.perf_counter()
    model = build_model(args.variant, args.nsrc, Wtot, local)
    host_setup_s = time.perf_counter() - t_setup

    def fence():
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    # the launches go on a non-blocking side stream (what a host application would use): the legacy
    # default stream serialises against every other stream of the process
    side = None if args.default_stream else torch.cuda.Stream(device=dev)
    if side is not None:
        torch.cuda.set_stream(side)

    def reduce_max(x):
        if not multi:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    leg = Leg(args, model, dev, local, world, rank, Wtot, shard)
    ctx, ndim, half = leg.ctx, leg.ndim, leg.half
    if args.profile_every <= 0:
        # (two evaluations per step.  One-launch evaluations: ONE run of --profile-span launches under one event pair in the timed
        # steps; forms of three launches: a pair around every steps-th evaluation's main kernel, two in all)
        args.profile_every = max(2 * args.steps if args.profile_span > 1 else args.steps, 1)
    args.profile_span = max(1, min(args.profile_span, args.profile_every))
    dt, kt, out = timed(leg, fence, args.warmup, args.steps, args.profile_level, agree=reduce_max, profile_every=args.profile_every,
                        profile_span=args.profile_span)
    dt = reduce_max(dt)
    assert torch.isfinite(out[0]).all() and torch.isfinite(out[1]).all(), "non-finite lnprob in the timed workload"
    step, theta_all, nblk = leg.step, leg.theta_all, leg.nblk

    evals = Wtot * args.steps
    value = evals / dt
    backend = "RCCL" if args.backend == "nccl" else args.backend
    if world == 1:
        par = "1 GPU"
    elif shard == "sources":
        par = "source-sharded x%d (1/%d of every field's sources and of the grid chunks per GPU), %s all-reduce of lnprob" % (world, world, backend)
    else:
        par = "walker-sharded x%d, %s all-gather of lnprob" % (world, backend)
        if leg.defer:
            par += " (--pipeline-collective: a block's gather overlaps the next block's kernel)"

    res = None
    if rank == 0:
        res = {"metric": "walker-lnprob evals/sec (%s sources, %d walkers%s)" % (
                   "10^%d" % round(np.log10(args.nsrc)) if 10 ** round(np.log10(args.nsrc)) == args.nsrc else str(args.nsrc),
                   W, " per GPU" if (scaling == "weak" and world > 1) else ""),
               "value": value,
               "unit": "walker-lnprob evals/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": scaling,
               "vs_baseline": None, "dtype": "f64", "data": "synthetic",
               "config": {"workload": "%s completeness, single Schechter: %d synthetic sources, %d walkers in all (%s), "
                                      "2 half-ensemble calls of %d theta rows per step"
                                      % (args.variant, args.nsrc, Wtot,
                                         "%d per GPU" % W if scaling == "weak" else "fixed ensemble", half),
                          "n_sources": args.nsrc, "walkers_per_gpu": Wtot // world if shard == "walkers" else Wtot,
                          "walkers_total": Wtot, "variant": args.variant, "shard": shard if world > 1 else "none",
                          "parallelism": par},
               "roofline": roofline_of(args, leg, model, kt, dt)}
    ab = [n for n in AB_FLAGS if getattr(args, n)]
    if ab and rank == 0:
        res["config"]["ab_options"] = ab                    # an A/B run, not the default path
    if rank == 0:
        if args.config:
            res["config"]["baseline_config"] = args.config
        # what precedes the first evaluation: the host setup of the class (cosmology tables, grids: NumPy / SciPy) and
        # lf_create (flux sort, cells, proven grid bins, uploads) - once per run, not in `value`
        res["setup_s"] = {"host_setup": host_setup_s, "lf_create": getattr(leg, "create_s", None),
                          "note": "once per catalogue, before the timed region"}
    if world > 1 and scaling == "weak" and not args.no_extras:
        # the metric's literal shape - a FIXED ensemble of --walkers walkers - on the same GPUs, in the same run:
        # sharded by walker (all-gather) and by source (all-reduce)
        strong = {}
        for sh in ("walkers", "sources"):
            try:
                lg = Leg(args, model, dev, local, world, rank, W, sh, seed=7)
                nst = max(10, args.steps)
                saved = args.steps
                args.steps = nst
                d2, kt2, o2 = timed(lg, fence, 2, nst, 1, agree=reduce_max)
                d2 = reduce_max(d2)
                if rank == 0:
                    strong[sh] = {"value": W * nst / d2, "unit": "walker-lnprob evals/s", "ms_per_step": d2 / nst * 1e3,
                                  "walkers_total": W, "rows_per_call_per_gpu": lg.rows_local, "sources_per_gpu": lg.ctx.N,
                                  "lf_main_ms": kt2["main"]["ms"] / max(kt2["main"]["launches"], 1)}
                args.steps = saved
                lg.close()
            except Exception as e:                 # an extra figure: never at the cost of the bench line
                strong[sh] = {"error": repr(e)}
        if rank == 0:
            res["strong_scaling"] = strong
    if rank == 0:
        if world == 1 and not args.no_extras:
            # the same workload as real MCMC: the device-resident sampler (theta, accept/reject and the
            # chain stay in HBM; six launches per ensemble step, no host in the loop)
            from lumfuncmcmc_amd.sampler import DeviceEnsembleSampler
            nst = max(10, args.steps)
            ds = DeviceEnsembleSampler(ctx, W, seed=1, capacity=nst + 3)
            ds.run_mcmc(theta_all[:2].reshape(-1, ndim)[:W], 3)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            ds.enqueue(None, nst)                      # the steps, on the device ...
            torch.cuda.synchronize()
            t2 = time.perf_counter() - t1
            ds.sync()                                  # ... and the chain's read-back, once per run, apart
            t3 = time.perf_counter() - t1 - t2
            res["mcmc_device_sampler"] = {"value": W * nst / t2, "unit": "walker-lnprob evals/s", "ms_per_step": t2 / nst * 1e3,
                                          "steps": nst, "chain_readback_ms": t3 * 1e3,
                                          "acceptance_fraction": float(ds.acceptance_fraction.mean()),
                                          "note": "counts every proposal, as emcee does; proposals outside the prior box are "
                                                  "-inf without being evaluated (MODE_SKIP)"}
            ds.close()
        if world == 1 and not args.no_extras:
            try:
                res["host"] = host_cost(leg)
                res["host_us_per_call"] = res["host"]["python_per_call"]["host_us"]
                # the timed workload again with its 2 x steps evaluations enqueued by ONE call of the C loop
                import gc
                K = 2 * args.steps
                ring = torch.stack([leg.blocks[j % nblk][leg.row_lo:leg.row_lo + leg.rows_local] for j in range(K)]).contiguous()
                out_n = torch.empty((K, leg.rows_local), dtype=torch.float64, device=dev)
                for _ in range(3):
                    ctx.lnprob_torch_n(ring, out=out_n)
                fence()
                gc.collect()
                gc.disable()
                t1 = time.perf_counter()
                ctx.lnprob_torch_n(ring, out=out_n)
                fence()
                t2 = time.perf_counter() - t1
                gc.enable()
                res["k_calls_per_c_call"] = {"value": W * args.steps / t2, "unit": "walker-lnprob evals/s", "ms_per_step": t2 / args.steps * 1e3,
                                             "evaluations_per_c_call": K,
                                             "note": "the same %d steps, their evaluations enqueued by one lf_lnprob_batch_device_n call "
                                                     "instead of %d calls from Python" % (args.steps, K)}
            except Exception as e:                 # an extra figure: never at the cost of the bench line
                res["host"] = {"error": repr(e)}
        if world == 1 and not args.no_extras and args.variant == "free" and not ab and leg.ctx.last_launch()["kind"] == 2:
            # The kernel north_star literally describes - every (walker, source) term evaluated, the catalogue streamed - is
            # what serves walkers that cannot be summed over the cells; timed here on the same workload with the cells off.
            try:
                ctx.set_option("cells", 0)
                nst = max(10, args.steps)
                saved = args.steps
                args.steps = nst
                d3, kt3, _ = timed(leg, fence, 2, nst, 1, profile_every=4)
                setattr(args, "no_cells", True)
                r3 = roofline_of(args, leg, model, kt3, d3)
                setattr(args, "no_cells", False)
                args.steps = saved
                ent = {}
                try:
                    ent = json.load(open(os.path.join(ROOT, "profiles", "hbm_traffic.json"))).get(
                        "free_n%d_b%d_nocells" % (leg.ctx.N, leg.rows_local), {})
                except Exception:
                    pass
                hb = ent.get("hbm_bytes_per_launch")
                res["per_source_kernel"] = {
                    "value": W * nst / d3, "unit": "walker-lnprob evals/s", "ms_per_step": d3 / nst * 1e3, "steps": nst,
                    "kernel": r3["kernel"], "avg_launch_ms": r3["avg_launch_ms"], "launches": r3["launches"],
                    "executed_tflops": r3["achieved"], "frac_of_fp64_peak": r3["frac"],
                    "items_per_launch_by_form": r3["items_per_launch_by_form"],
                    "algorithmic_bytes_per_launch": r3["hbm"]["algorithmic_bytes_per_launch"],
                    "algorithmic_gb_per_s": r3["hbm"]["achieved"], "hbm_peak_gb_per_s": HBM_PEAK_GBS,
                    "hbm_bytes_per_launch_pmc": hb,
                    "hbm_gb_per_s_pmc": (hb / (r3["avg_launch_ms"] * 1e-3) / 1e9) if hb else None,
                    "note": "option cells = 0: every walker over the sources (lf_free's table-driven per-source path: "
                            "the catalogue's log-flux streamed by 8-byte loads, 8 B per source and launch); compute-bound on fp64 "
                            "VALU issue by design - the HBM rate is a few per cent of the peak (pmc figures: the committed "
                            "rocprofv3 passes in profiles/hbm_traffic.json)"}
            except Exception as e:
                res["per_source_kernel"] = {"error": repr(e)}
            finally:
                ctx.set_option("cells", 1)
        if world == 1 and not args.compress and not args.no_extras:
            # separately labelled, NOT the headline: the same workload with piece A taken from the compressed
            # catalogue (opt-in "compress" option, csrc/lf_compress.h) - the roofline above is the direct kernel's
            res["compressed_catalogue"] = compressed_leg(ctx, step, fence, out, args.steps, W)
        if world == 1 and not args.no_cpu_baseline:
            cb, ref = cpu_baseline(model, args.variant, theta_all[(2 * (args.steps - 1)) % nblk], args.cpu_budget)
            got = out[0].cpu().numpy()[:len(ref)]
            cb["max_rel_diff_gpu_vs_port"] = float(np.max(np.abs(got - ref) / np.abs(ref)))
            cb["host_cpu_count"] = os.cpu_count()
            res["cpu_baseline"] = cb
            try:
                # the GPU box gives one GPU's share of the host: 16 cores (more threads only oversubscribe the quota)
                nthr = int(os.environ.get("LF_BENCH_THREADS", min(len(os.sched_getaffinity(0)), 16)))
                res["cpu_baseline_allcores"] = cpu_baseline_allcores(model, theta_all[0], nthr)
                res["cpu_baseline_allcores"]["host_cpu_count"] = os.cpu_count()
            except Exception as e:            # the extra figure must never cost the bench line
                res["cpu_baseline_allcores"] = {"error": str(e)}
        os.write(json_fd, (json.dumps(res) + "\n").encode())
    leg.close()
    if multi:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
