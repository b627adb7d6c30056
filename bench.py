#!/usr/bin/env python3
"""Headline benchmark: MLUPS of the D3Q19x2 binary fluctuating-LBM step on MI355X.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--size S] [--noise] [--schedule fused|two_pass] [--transport T]

A "step" is one LBM_timestep-equivalent (stream + densities + noise + projection + collide,
LBM_binary.H:545-594) over the whole lattice.  N=1: the headline `value` is the 512^3 periodic box at zero
noise (stripe init) that BASELINE.json quotes its 60 % target on; configs[1]'s 256^3 is timed in the same run
and reported, with its own roofline, under config.also.  N>1: weak scaling, every GPU owns a 512x512x512
z-slab of a 512x512x(512 N) box, +-z planes exchanged and overlapped with the interior planes; at N=4 / N=8 the slab
shapes of configs[3] (512x512x128, droplet) / configs[4] (1024x1024x64, mixture with kBT > 0, alpha0 = 2.5) are timed in the same
run under config.also.  --size S / --shape NX,NY,NZ time one explicit case instead.
Populations are resident in HBM before the timed region.  Prints ONE JSON line on rank 0.

N > 1, who does what (the reference distributes boxes over MPI ranks, main_run_job.cpp:140-143, and fills ghost cells
with FillBoundary, LBM_binary.H:553-555):
  * launched by hand (`python bench.py --gpus N`): starts `torch.distributed.run` with N ranks of this file;
  * a rank started by torch.distributed.run is a SUPERVISOR: it never touches the GPU.  The supervisors form a gloo
    group, and for each halo transport of the chain (--transport auto: rccl -> peer-copy -> peer-kernel -> rccl-direct)
    they start FRESH worker processes on a fresh rendezvous port, watch them under a wall-clock limit, agree on the
    outcome, and kill the workers (their own process groups) when one failed or the limit passed.  The first transport
    that completes produces the line; config.launcher says which ones were tried.  An RCCL failure is usually a hang or
    an abort in one rank, which a retry inside the ranks cannot survive -- hence fresh processes.  --second-transport also
    times the next transport family beside the first (config.second_transport; off by default);
  * a WORKER (BFLBM_BENCH_WORKER=1) does the measurement: transports `rccl` (pack, one message per face, unpack) and
    `rccl-direct` (38 plane-sized sends per face straight from the state) run one process per GPU over
    torch.distributed; `peer-kernel` / `peer-copy` run ONE process that drives all N GPUs through the C-ABI's ring
    (bflbm_ring_*): faces moved by a gather kernel reading peer memory in place, or by the copy engine
    (hipMemcpyPeerAsync: no compute units, so nothing competes with the one-workgroup-per-CU interior sweep).
"""
import argparse
import json
import os
import signal
import socket
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

METRIC = "MLUPS (million lattice-site updates/sec) D3Q19×2 at 256³/512³; % HBM roofline"   # BASELINE.json
BYTES_PER_LUP = 608.0          # 2 fluids x 19 populations x 8 B x (1 read + 1 write), BASELINE.md section 3
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md
REFERENCE_PROBE_MLUPS = 0.38   # the reference's own headers behind a container shim, 1 core (survey host)

TRANSPORT_CHAIN = ["rccl", "peer-copy", "peer-kernel", "rccl-direct"]   # after RCCL the most standard multi-GPU operation HIP has (peer copies), then peer reads by a kernel
TRANSPORT_TEXT = {
    "rccl": "rccl, staged (pack, one message per face, unpack)",
    "rccl-direct": "rccl, direct (38 plane-sized sends per face from the state buffers)",
    "peer-kernel": "peer, one process drives all GPUs; a gather kernel reads the neighbour's planes in place (xGMI peer memory)",
    "peer-copy": "peer, one process drives all GPUs; copy engine, 38 hipMemcpyPeerAsync per face (no compute units)",
}


def cpu_baseline(noise=False):
    """Time the CPU oracle (restated reference path) on a bounded sample: 1 thread = the reference as
    shipped (serial build, GNUmakefile:16-19) is the reported value; the same code threaded over z planes
    on all host cores is added for information.  At kBT = 0 the oracle skips the generator (the noise is
    exactly 0) while the reference still draws its 33 normals per site (LBM_binary.H:113-127; 48 % of its time,
    SURVEY section 6), so the kBT > 0 figure -- generator included -- is reported beside it (`with_noise`) and is
    the reported value of a --noise run: like with like.  Both settings run the same number of steps.  The port is
    faster than the reference as shipped (no ghost-cell fill, no dead update_com): the survey's own probe of the
    reference headers is carried along as `reference_probe_mlups`."""
    import oracle_binding as ob
    n, steps = 64, 24
    ob.lib().orc_set_threads(1)
    secs, _ = ob.bench(n, n, n, steps)
    pn = ob.default_params()
    pn.kBT, pn.alpha0 = 1e-5, 0.0
    secs_n, _ = ob.bench(n, n, n, steps, params=pn)
    quiet = round(n ** 3 * steps / secs / 1e6, 4)
    noisy = round(n ** 3 * steps / secs_n / 1e6, 4)
    out = {"value": noisy if noise else quiet, "unit": "MLUPS", "cores": 1, "kind": "port",
           "sample": (f"{n}^3 stripe, kBT=1e-5 (33 normals per site drawn), {steps} steps" if noise else
                      f"{n}^3 stripe, kBT=0 (generator skipped), {steps} steps") +
                     " after 1 warm-up, oracle/bflbm_oracle.c -O3 -ffp-contract=off, 1 thread",
           "zero_noise": quiet, "with_noise": noisy,
           "reference_probe_mlups": REFERENCE_PROBE_MLUPS,
           "reference_probe": "BASELINE.md section 2: the reference's LBM_binary.H behind a container shim, 1 core of the survey host, "
                              "with its ghost-cell fills, dead update_com and the generator at kBT = 0",
           "host_cores": os.cpu_count()}
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = min(cores, 16)             # a one-GPU box is given a 16-core share of its host, whatever it reports
    if cores > 1:
        n2, steps2 = 128, 20
        ob.lib().orc_set_threads(cores)
        secs2, _ = ob.bench(n2, n2, n2, steps2)
        ob.lib().orc_set_threads(1)
        out["all_cores"] = {"value": round(n2 ** 3 * steps2 / secs2 / 1e6, 3), "unit": "MLUPS", "cores": cores,
                            "sample": f"{n2}^3 stripe, kBT=0, {steps2} steps, same code with OpenMP over z planes"}
    return out


def load_traffic(workload, schedule, field="hbm_bytes_per_launch"):
    """From the committed rocprofv3 passes of this workload (profiles/traffic.json, written by tools/make_profiles.py):
    HBM bytes per launch (--pmc FETCH_SIZE x2 + WRITE_SIZE) or, field="kernel_avg_ms", the average launch duration
    of the step's kernel(s) in the --kernel-trace --stats summary; null when the workload has not been profiled."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        with open(path) as fh:
            t = json.load(fh)
        e = t.get(f"{workload}|{schedule}")
        return None if e is None else e.get(field)
    except Exception:
        return None


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--size", type=int, default=0, help="cubic box edge at N=1 / slab edge per GPU (default: 512, with 256 reported beside it at N=1)")
    ap.add_argument("--shape", default="", help="NX,NY,NZ per GPU instead of a cube (e.g. 1024,1024,64 = configs[4]'s slab)")
    ap.add_argument("--noise", action="store_true", help="kBT=1e-5 (configs[2]) instead of zero noise")
    ap.add_argument("--init", default=None, choices=["stripe", "droplet", "mixture"],
                    help="initial state; default: stripe at zero noise, mixture with --noise (configs[2] = NoiseCovariance.ipynb's homogeneous mixture)")
    ap.add_argument("--schedule", default=os.environ.get("BFLBM_SCHEDULE", "auto"))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--blocks", type=int, default=3, help="the K-step block is timed this many times; value = the median block")
    ap.add_argument("--transport", default=os.environ.get("BFLBM_BENCH_TRANSPORT_CHAIN", "auto"),
                    help="N > 1: auto (the chain " + " -> ".join(TRANSPORT_CHAIN) + ", first that completes), one of them, or a comma list")
    ap.add_argument("--attempt-timeout", type=float, default=420.0, help="N > 1: wall-clock limit of one transport attempt, seconds")
    ap.add_argument("--no-overlap-leg", action="store_true", help="N > 1: skip the informational steps with the exchange after the sweep")
    ap.add_argument("--second-transport", action="store_true", default=os.environ.get("BFLBM_BENCH_SECOND_TRANSPORT", "0") == "1",
                    help="N > 1, --transport auto: after the first transport completed, time the next transport FAMILY as an informational "
                         "leg (config.second_transport).  Off by default: a transport that has never run on several GPUs should not get "
                         "the chance to disturb the node in the middle of a scaling series once a result exists")
    ap.add_argument("--no-second-transport", action="store_true", help="(kept for older command lines; the leg is off unless --second-transport)")
    return ap.parse_args()


# ---------------------------------------------------------------------------------------------------------------------
# N > 1: launcher and supervisors (no GPU is touched in this part of the file)

def _free_port():
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def launch_ranks(a):
    """Started by hand without a launcher: one supervisor rank per GPU under torch.distributed.run; their exit code is ours."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    raise SystemExit(subprocess.run(cmd).returncode)


def _kill_group(child):
    """End the worker we started and whatever it started: its own session / process group, by its exact id."""
    if child is None or child.poll() is not None:
        return
    for sig, wait in ((signal.SIGTERM, 5.0), (signal.SIGKILL, 5.0)):
        try:
            os.killpg(child.pid, sig)
        except ProcessLookupError:
            return
        t0 = time.time()
        while time.time() - t0 < wait:
            if child.poll() is not None:
                return
            time.sleep(0.1)


def supervise(a):
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    if a.gpus != world:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    dist.init_process_group("gloo")                    # control plane of the supervisors: CPU only, torchrun's rendezvous
    chain = TRANSPORT_CHAIN if a.transport == "auto" else [t.strip() for t in a.transport.split(",") if t.strip()]
    for t in chain:
        if t not in TRANSPORT_TEXT:
            raise SystemExit(f"unknown transport {t!r}; known: {', '.join(TRANSPORT_TEXT)}")
    t_begin = time.time()

    def attempt(transport, limit):
        """Start fresh workers for one transport, watch them, agree on the outcome.  Returns (ok, line or None, note)."""
        port = [_free_port() if rank == 0 else 0]
        dist.broadcast_object_list(port, src=0)
        single = transport.startswith("peer")          # one process drives all GPUs
        mine = (rank == 0) or not single
        child, outf = None, None
        if mine:
            env = dict(os.environ, BFLBM_BENCH_WORKER="1", BFLBM_BENCH_TRANSPORT=transport, MASTER_ADDR="127.0.0.1",
                       MASTER_PORT=str(port[0]), TORCHELASTIC_USE_AGENT_STORE="False")
            outf = tempfile.TemporaryFile(mode="w+")
            child = subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                     stdout=outf, start_new_session=True)
        t0 = time.time()
        note = ""
        while True:
            rc = child.poll() if child is not None else 0
            late = rank == 0 and time.time() - t0 > limit
            st = torch.tensor([1.0 if rc == 0 else 0.0, 1.0 if (rc not in (None, 0) or late) else 0.0, 1.0 if late else 0.0])
            dist.all_reduce(st)
            n_ok, n_bad, n_late = (int(v) for v in st.tolist())
            if n_bad > 0:
                _kill_group(child)
                note = f"limit of {limit:.0f} s passed" if n_late else "a worker exited with an error"
                ok = False
                break
            if n_ok == world:
                ok = True
                break
            time.sleep(0.5)
        line = None
        if ok and rank == 0:
            outf.seek(0)
            js = [l for l in outf.read().splitlines() if l.startswith("{")]
            if js:
                line = json.loads(js[-1])
            else:
                ok, note = False, "the workers finished without a result line"
        flag = [ok]
        dist.broadcast_object_list(flag, src=0)
        if outf is not None:
            outf.close()
        return flag[0], line, note, round(time.time() - t0, 1)

    tried, result, done = [], None, False              # result: the line (rank 0 only); done: a transport completed (all ranks)
    for k, transport in enumerate(chain):
        ok, line, note, secs = attempt(transport, a.attempt_timeout)
        tried.append({"transport": transport, "ok": bool(ok), "seconds": secs, **({"note": note} if note else {})})
        if rank == 0:
            print(f"[bench] transport {transport}: {'ok' if ok else 'FAILED (' + note + ')'} after {secs} s", file=sys.stderr, flush=True)
        if ok:
            result, done = line, True
            break
    # Informational: the next transport FAMILY (peer after rccl, rccl after peer) in the same run, when the first attempt left
    # time for it -- the driver's scaling run is the only multi-GPU measurement a round gets.  Never allowed to cost the line.
    second = None
    if done and a.transport == "auto" and a.second_transport and not a.no_second_transport:
        fam = tried[-1]["transport"].split("-")[0]
        alt = next((t for t in TRANSPORT_CHAIN if t.split("-")[0] != fam and t not in [x["transport"] for x in tried]), None)
        go = [alt is not None and time.time() - t_begin < 200.0]
        dist.broadcast_object_list(go, src=0)
        if go[0]:
            ok, line, note, secs = attempt(alt, min(a.attempt_timeout, 240.0))
            tried.append({"transport": alt, "ok": bool(ok), "seconds": secs, "informational": True, **({"note": note} if note else {})})
            if ok and rank == 0:
                second = {"halo_transport": line["config"]["halo_transport"], "value": line["value"], "ms_per_step": line["ms_per_step"],
                          "spread": line.get("spread"), "halo_overlap": line["config"].get("halo_overlap"),
                          "roofline_frac": line["roofline"]["frac"], "also": line["config"].get("also")}
    if rank == 0 and result is not None:
        result["config"]["launcher"] = {"transports_tried": tried, "chain": chain,
                                        "note": "each attempt = fresh worker processes on a fresh rendezvous port, watched under a wall-clock limit"}
        if second is not None:
            result["config"]["second_transport"] = second
        print(json.dumps(result), flush=True)
    dist.barrier()
    dist.destroy_process_group()
    if not done and rank == 0:
        print(f"[bench] no transport completed: {tried}", file=sys.stderr, flush=True)
    raise SystemExit(0 if done else 1)


# ---------------------------------------------------------------------------------------------------------------------
# the measurement (single GPU, or a worker of an N > 1 run)

def work(a):
    # Libraries print banners on the C-level stdout (RCCL: "Librccl path : ..."); the contract is ONE JSON line
    # on stdout, so everything until that line goes to stderr.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    worker = os.environ.get("BFLBM_BENCH_WORKER") == "1"
    transport = os.environ.get("BFLBM_BENCH_TRANSPORT", "rccl") if worker else None
    world = int(os.environ.get("WORLD_SIZE", "1")) if worker else 1
    rank = int(os.environ.get("RANK", "0")) if worker else 0
    local_rank = int(os.environ.get("LOCAL_RANK", "0")) if worker else 0
    single_process_ring = worker and transport.startswith("peer")

    # test hooks of the launcher (tests/test_bench_launcher.py, tests/test_gpu_slabs.py): make one transport's workers fail
    # the way RCCL does (one rank exits, or one rank never arrives), or replace the measurement by a rendezvous check
    fail = os.environ.get("BFLBM_BENCH_FAIL", "")
    for item in fail.split(","):
        if worker and item and item.split(":")[0] == transport and rank == (0 if single_process_ring else world - 1):
            if item.split(":")[1] == "exit":
                raise SystemExit(7)
            time.sleep(1e6)
    if worker and os.environ.get("BFLBM_BENCH_FAKE_WORKER") == "1":
        import torch
        import torch.distributed as dist
        if not single_process_ring:
            dist.init_process_group("gloo")
            t = torch.ones(1)
            dist.all_reduce(t)
            assert int(t.item()) == world
            dist.destroy_process_group()
        if rank == 0:
            os.dup2(real_stdout, 1)
            print(json.dumps({"metric": METRIC, "value": 0.0, "unit": "MLUPS", "n_gpus": world, "ms_per_step": 0.0, "data": "none: launcher test hook",
                              "config": {"halo_transport": TRANSPORT_TEXT[transport]}, "roofline": {"frac": 0.0}}), flush=True)
        return

    import __graft_entry__ as ge
    pkg = ge.load_package()

    use_dist = worker and not single_process_ring
    if use_dist:
        import torch
        import torch.distributed as dist
        # one rank per GPU over RCCL; BFLBM_BENCH_BACKEND=gloo rehearses several ranks on a box with fewer
        # GPUs (ranks then share devices and the halo goes through the host) -- never a measurement
        backend = os.environ.get("BFLBM_BENCH_BACKEND", "nccl")
        local_rank = local_rank % max(torch.cuda.device_count(), 1)
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            # RCCL's copy kernels on a high-priority stream: the interior sweep holds every CU with one 512-register
            # workgroup, and the faces should take the first CU a finishing workgroup frees, not queue behind the next round
            try:
                opts = dist.ProcessGroupNCCL.Options(is_high_priority_stream=True)
            except Exception:                      # noqa: BLE001 -- an older torch without the option: default priority
                opts = None
            kw = {} if opts is None else {"pg_options": opts}
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank), **kw)
        else:
            dist.init_process_group(backend)
        os.environ["BFLBM_SLAB_TRANSPORT"] = "direct" if transport == "rccl-direct" else "staged"
    ring_devices = None
    if single_process_ring:
        n = ctypes_device_count(pkg)
        ring_devices = list(range(min(world, max(n, 1))))

    def run_case(sx, sy, sz, init=None, noise=None, alpha0=None):
        """Time a.steps steps on a lattice of sx x sy x (sz*world); returns the result fields of this case (rank 0)."""
        noise = a.noise if noise is None else noise
        # default initial state: the stripe at zero noise; with noise the homogeneous mixture of NoiseCovariance.ipynb (configs[2]).
        # It matters for the time: on a state that the noise has randomised a step takes 2-3 % longer than on the smooth stripe
        # (the same kernel, profiles/r04_sustained.txt), so a noisy run timed from a stripe looks faster for its first hundred steps
        init = (a.init or ("mixture" if noise else "stripe")) if init is None else init
        par = dict(kBT=1e-5, alpha0=0.0 if alpha0 is None else alpha0) if noise else ({} if alpha0 is None else dict(alpha0=alpha0))
        params = pkg.default_params(**par)
        init_args = [0.5] if init == "stripe" else [0.2] if init == "droplet" else []
        nx, ny, nz = sx, sy, sz * world
        if use_dist:
            def barrier():
                torch.cuda.synchronize()
                dist.barrier()
                torch.cuda.synchronize()
            lat = pkg.SlabLattice(nx, ny, nz, params=params, schedule=a.schedule)
            eng = lat.engine
        elif single_process_ring:
            lat = pkg.RingLBM(nx, ny, nz, nslabs=world, devices=ring_devices, params=params, schedule=a.schedule)
            lat.set_transport("copy" if transport == "peer-copy" else "kernel")
            eng = lat.slabs[0]

            def barrier():
                lat.sync()
        else:
            lat = pkg.BinaryLBM(nx, ny, nz, params=params, device=local_rank, schedule=a.schedule)
            eng = lat

            def barrier():
                eng.sync()
        getattr(lat, "LBM_init_" + init)(*init_args)
        lat.LBM_timestep(a.warmup)
        barrier()

        eng_schedule = eng.resolved_schedule() if hasattr(eng, "resolved_schedule") else a.schedule    # what auto resolves to
        placement = eng.placement_report() if hasattr(eng, "placement_report") else None
        # EXACTLY a.steps steps per timed block, each block bracketed by barrier + device synchronisation on both sides and
        # reduced with MAX over the ranks; the block is repeated a.blocks times and the MEDIAN block is the reported one
        # (boxes of the pool, and runs on one box, scatter by a few percent: `spread` shows the blocks).
        blocks = []
        for _ in range(max(1, a.blocks)):
            barrier()
            if not single_process_ring:
                eng.timer_start()
            t0 = time.perf_counter()
            lat.LBM_timestep(a.steps)
            dev_ms = eng.timer_stop() if not single_process_ring else None   # hipEvents on the stream the kernels run on
            barrier()
            wall = time.perf_counter() - t0
            if dev_ms is None:
                dev_ms = wall * 1e3                # N GPUs driven by one process: the block's wall clock between device synchronisations
            if use_dist:
                t = torch.tensor([wall, dev_ms], dtype=torch.float64, device="cuda")
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                wall, dev_ms = t.tolist()
            blocks.append((wall, dev_ms))
        order = sorted(range(len(blocks)), key=lambda i: blocks[i][0])
        wall, dev_ms = blocks[order[len(order) // 2]]
        spread = {"blocks_ms_per_step": [round(b[0] / a.steps * 1e3, 4) for b in blocks],
                  "min": round(min(b[0] for b in blocks) / a.steps * 1e3, 4), "max": round(max(b[0] for b in blocks) / a.steps * 1e3, 4),
                  "reported": "median block"}
        # N > 1: the same steps with the exchange AFTER the sweep instead of behind it (configs[3]: "overlap
        # efficiency"); outside the timed region, and never allowed to break the headline line
        seq_ms = None
        if world > 1 and not a.no_overlap_leg:
            try:
                if use_dist:
                    lat.overlap = False
                else:
                    lat.set_overlap(False)
                nseq = max(2, min(a.steps, 20))
                lat.LBM_timestep(2)
                barrier()
                t1 = time.perf_counter()
                lat.LBM_timestep(nseq)
                barrier()
                seq_ms = (time.perf_counter() - t1) / nseq * 1e3
                if use_dist:
                    seq = torch.tensor([seq_ms], dtype=torch.float64, device="cuda")
                    dist.all_reduce(seq, op=dist.ReduceOp.MAX)
                    seq_ms = float(seq.item())
            except Exception as exc:               # noqa: BLE001 -- informational leg only
                seq_ms = None
                print(f"[bench] sequential-exchange leg skipped: {exc}", file=sys.stderr)
            finally:
                if use_dist:
                    lat.overlap = True
                else:
                    lat.set_overlap(True)
        rho_sum, phi_sum = lat.mass()
        if use_dist:
            halo_bytes = lat.halo_bytes_per_face
            tr_text = TRANSPORT_TEXT["rccl-direct" if lat.direct else "rccl"] + ("" if dist.get_backend() == "nccl" else f" [rehearsal over {dist.get_backend()}]")
        elif single_process_ring:
            halo_bytes = int(lat.slabs[0].halo_bytes(pkg._lib.HALO_STATE))
            kf, cf = lat.last_transport()
            tr_text = TRANSPORT_TEXT[transport] + f" [{kf} faces by kernel, {cf} by copies, on {len(ring_devices)} device(s)]"
        else:
            halo_bytes, tr_text = None, None
        if use_dist:
            dist.barrier()
        lat.close()
        sites = float(nx) * ny * nz
        per_gpu_sites = sites / world
        kern_ms = dev_ms / a.steps
        achieved = per_gpu_sites * BYTES_PER_LUP / (kern_ms * 1e-3) / 1e9
        schedule = eng_schedule
        workload = f"{nx}x{ny}x{nz} periodic, {init} init, " + (f"kBT=1e-5 alpha0={params.alpha0:g}" if noise else "zero noise")
        return {
            "value": round(sites * a.steps / wall / 1e6, 1), "ms_per_step": round(wall / a.steps * 1e3, 4),
            "workload": workload, "schedule": schedule, "slab_per_gpu": f"{nx}x{ny}x{nz // world}", "spread": spread,
            "mass_check": [rho_sum, phi_sum], "halo_bytes_per_face": halo_bytes, "halo_transport": tr_text, "placement": placement,
            "halo_overlap": None if seq_ms is None else {"ms_per_step_overlapped": round(wall / a.steps * 1e3, 4),
                                                          "ms_per_step_exchange_after_sweep": round(seq_ms, 4)},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4),
                         "traffic": load_traffic(f"{sx}x{sy}x{sz}" + (" noise" if noise else ""), schedule),
                         "kernel": "all kernels of one step (hipEvent time / steps)" if not single_process_ring else
                                   "all kernels of one step on every GPU (wall clock between device synchronisations / steps)",
                         "algorithmic_bytes_per_launch": per_gpu_sites * BYTES_PER_LUP,
                         "avg_launch_ms": round(kern_ms, 4),
                         "kernel_ms_from_profiles": load_traffic(f"{sx}x{sy}x{sz}" + (" noise" if noise else ""), schedule, "kernel_avg_ms")},
        }

    # Workload.  The 60 % target of BASELINE.json is quoted on a 512^3 step at 1 GPU, so that is the headline
    # (81.6 GB of populations); configs[1]'s 256^3 is timed in the same run and reported under config.also.
    # N > 1: weak scaling with the same 512x512x512 slab per GPU, and the slab shapes of configs[3] / configs[4] at the GPU
    # counts they are quoted on.  --size / --shape select one explicit case.
    explicit = bool(a.shape) or a.size > 0
    if a.shape:
        head = tuple(int(v) for v in a.shape.split(","))
    else:
        S = a.size if a.size > 0 else 512
        head = (S, S, S)
    also = {}

    def add_also(r):
        also[r["workload"]] = {k: r[k] for k in ("value", "ms_per_step", "spread", "schedule", "roofline", "halo_overlap", "slab_per_gpu", "placement") if r.get(k) is not None}
    if not explicit and world == 1:
        add_also(run_case(256, 256, 256))
    res = run_case(*head)
    if not explicit and world > 1:
        try:                                           # never allowed to cost the headline line
            if world == 4:
                add_also(run_case(512, 512, 128, init="droplet", noise=False))                # configs[3]: 512^3 droplet on 4 z-slabs
            if world == 8:
                # configs[4]: 1024x1024x512 spinodal mixture.  alpha0 = 2.5: SURVEY 8d's example alpha0 = 4 on the rho = phi = 1 mixture
                # is NaN within 50 steps on the reference's own CPU path (tests/test_oracle_pins.py); 2.5 demixes and stays finite
                add_also(run_case(1024, 1024, 64, init="mixture", noise=True, alpha0=2.5))
        except Exception as exc:                       # noqa: BLE001
            print(f"[bench] config.also case skipped: {exc!r}", file=sys.stderr)

    if rank == 0:
        out = {
            "metric": METRIC, "value": res["value"], "unit": "MLUPS",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": res["ms_per_step"],
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "spread": res["spread"],
            "config": {"workload": res["workload"], "schedule": res["schedule"], "slab_per_gpu": res["slab_per_gpu"],
                       "parallelism": f"z-slab x{world}" if world > 1 else "single GPU",
                       "mass_check": res["mass_check"], "halo_bytes_per_face": res["halo_bytes_per_face"],
                       "halo_transport": res["halo_transport"],
                       "halo_overlap": res["halo_overlap"], "placement": res["placement"], "also": also or None},
            "roofline": res["roofline"],
        }
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(noise=a.noise)
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)
    if use_dist:
        dist.destroy_process_group()


def ctypes_device_count(pkg):
    import ctypes
    n = ctypes.c_int()
    if pkg._lib.load().bflbm_device_count(ctypes.byref(n)) != 0:
        return 1
    return n.value


def main():
    a = parse_args()
    if os.environ.get("BFLBM_BENCH_WORKER") == "1" or a.gpus <= 1:
        return work(a)
    if "RANK" in os.environ and "MASTER_ADDR" in os.environ:
        return supervise(a)                            # a rank of torch.distributed.run
    return launch_ranks(a)


if __name__ == "__main__":
    main()
