#!/usr/bin/env python3
"""Headline benchmark: MLUPS of the D3Q19x2 binary fluctuating-LBM step on MI355X.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--size S] [--noise] [--schedule fused|two_pass]

A "step" is one LBM_timestep-equivalent (stream + densities + noise + projection + collide,
LBM_binary.H:545-594) over the whole lattice.  N=1: the headline `value` is the 512^3 periodic box at zero
noise (stripe init) that BASELINE.json quotes its 60 % target on; configs[1]'s 256^3 is timed in the same run
and reported, with its own roofline, under config.also.  N>1: weak scaling, every GPU owns a 512x512x512
z-slab of a 512x512x(512 N) box, +-z planes exchanged over RCCL and overlapped with the interior planes.
--size S / --shape NX,NY,NZ time one explicit case instead.
Populations are resident in HBM before the timed region.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

METRIC = "MLUPS (million lattice-site updates/sec) D3Q19\u00d72 at 256\u00b3/512\u00b3; % HBM roofline"   # BASELINE.json
BYTES_PER_LUP = 608.0          # 2 fluids x 19 populations x 8 B x (1 read + 1 write), BASELINE.md section 3
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md


def cpu_baseline(noise=False):
    """Time the CPU oracle (restated reference path) on a bounded sample: 1 thread = the reference as
    shipped (serial build, GNUmakefile:16-19) is the reported value; the same code threaded over z planes
    on all host cores is added for information.  At kBT = 0 the oracle skips the generator (the noise is
    exactly 0) while the reference still draws its 33 normals per site (LBM_binary.H:113-127; 48 % of its time,
    SURVEY section 6), so the kBT > 0 figure -- generator included -- is reported beside it (`with_noise`) and is
    the reported value of a --noise run: like with like."""
    import oracle_binding as ob
    n, steps = 64, 40
    ob.lib().orc_set_threads(1)
    secs, _ = ob.bench(n, n, n, steps)
    pn = ob.default_params()
    pn.kBT, pn.alpha0 = 1e-5, 0.0
    nsteps_n = 12
    secs_n, _ = ob.bench(n, n, n, nsteps_n, params=pn)
    quiet = round(n ** 3 * steps / secs / 1e6, 4)
    noisy = round(n ** 3 * nsteps_n / secs_n / 1e6, 4)
    out = {"value": noisy if noise else quiet, "unit": "MLUPS", "cores": 1, "kind": "port",
           "sample": (f"{n}^3 stripe, kBT=1e-5 (33 normals per site drawn), {nsteps_n} steps" if noise else
                      f"{n}^3 stripe, kBT=0 (generator skipped), {steps} steps") +
                     " after 1 warm-up, oracle/bflbm_oracle.c -O3 -ffp-contract=off, 1 thread",
           "zero_noise": quiet, "with_noise": noisy,
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--size", type=int, default=0, help="cubic box edge at N=1 / slab edge per GPU (default: 512, with 256 reported beside it at N=1)")
    ap.add_argument("--shape", default="", help="NX,NY,NZ per GPU instead of a cube (e.g. 1024,1024,64 = configs[4]'s slab)")
    ap.add_argument("--noise", action="store_true", help="kBT=1e-5 (configs[2]) instead of zero noise")
    ap.add_argument("--init", default="stripe", choices=["stripe", "droplet", "mixture"])
    ap.add_argument("--schedule", default=os.environ.get("BFLBM_SCHEDULE", "auto"))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--blocks", type=int, default=3, help="the K-step block is timed this many times; value = the median block")
    a = ap.parse_args()

    if a.gpus > 1 and "RANK" not in os.environ:
        # started by hand without a launcher: start one rank per GPU as child processes (this process has not
        # touched the GPU) and hand their exit code on
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.run(cmd).returncode)

    # Libraries print banners on the C-level stdout (RCCL: "Librccl path : ..."); the contract is ONE JSON line
    # on stdout, so everything until that line goes to stderr.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    import __graft_entry__ as ge
    pkg = ge.load_package()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus != world and world > 1:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    par = dict(kBT=1e-5, alpha0=0.0) if a.noise else {}
    params = pkg.default_params(**par)

    use_dist = "RANK" in os.environ and "MASTER_ADDR" in os.environ     # launched by torch.distributed.run
    if use_dist:
        import torch
        import torch.distributed as dist
        # one rank per GPU over RCCL; BFLBM_BENCH_BACKEND=gloo rehearses several ranks on a box with fewer
        # GPUs (ranks then share devices and the halo goes through the host) -- never a measurement
        backend = os.environ.get("BFLBM_BENCH_BACKEND", "nccl")
        local_rank = local_rank % max(torch.cuda.device_count(), 1)
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    def run_case(sx, sy, sz):
        """Time a.steps steps on a lattice of sx x sy x (sz*world); returns the result fields of this case (rank 0)."""
        nx, ny, nz = sx, sy, sz * world
        transport = None
        if use_dist:
            def barrier():
                torch.cuda.synchronize()
                dist.barrier()
                torch.cuda.synchronize()

            def make_and_warm():
                lat_ = pkg.SlabLattice(nx, ny, nz, params=params, schedule=a.schedule)
                try:
                    getattr(lat_, "LBM_init_" + a.init)(*([0.5] if a.init == "stripe" else [0.2] if a.init == "droplet" else []))
                    lat_.LBM_timestep(a.warmup)
                    barrier()
                except Exception:
                    lat_.close()                   # 87 GB per rank at the default size
                    raise
                return lat_
            try:
                lat = make_and_warm()
            except Exception as exc:               # noqa: BLE001
                # the staging-free transport (plane-sized sends between the state buffers) has only met gloo so far; if
                # RCCL refuses it, fall back to one packed message per face rather than lose the measurement -- all
                # ranks see the same error class at the same call, and the line says which transport ran
                print(f"[bench] rank {rank}: direct halo transport failed ({exc!r}); retrying staged", file=sys.stderr)
                os.environ["BFLBM_SLAB_STAGED"] = "1"
                lat = make_and_warm()
            transport = "direct (38 plane-sized sends per face from the state buffers)" if lat.direct else "staged (pack, one message per face, unpack)"
            eng = lat.engine
        else:
            lat = pkg.BinaryLBM(nx, ny, nz, params=params, device=local_rank, schedule=a.schedule)
            eng = lat

            def barrier():
                eng.sync()
            getattr(lat, "LBM_init_" + a.init)(*([0.5] if a.init == "stripe" else [0.2] if a.init == "droplet" else []))
            lat.LBM_timestep(a.warmup)

        eng_schedule = eng.resolved_schedule() if hasattr(eng, "resolved_schedule") else a.schedule    # what auto resolves to
        # EXACTLY a.steps steps per timed block, each block bracketed by barrier + device synchronisation on both sides and
        # reduced with MAX over the ranks; the block is repeated a.blocks times and the MEDIAN block is the reported one
        # (boxes of the pool, and runs on one box, scatter by a few percent: `spread` shows the blocks).
        blocks = []
        for _ in range(max(1, a.blocks)):
            barrier()
            eng.timer_start()
            t0 = time.perf_counter()
            lat.LBM_timestep(a.steps)
            dev_ms = eng.timer_stop()              # hipEvents on the stream the kernels run on
            barrier()
            wall = time.perf_counter() - t0
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
        if use_dist and world > 1:
            try:
                lat.overlap = False
                nseq = max(2, min(a.steps, 20))
                lat.LBM_timestep(2)
                barrier()
                t1 = time.perf_counter()
                lat.LBM_timestep(nseq)
                barrier()
                seq = torch.tensor([(time.perf_counter() - t1) / nseq * 1e3], dtype=torch.float64, device="cuda")
                dist.all_reduce(seq, op=dist.ReduceOp.MAX)
                seq_ms = float(seq.item())
            except Exception as exc:               # noqa: BLE001 -- informational leg only
                seq_ms = None
                print(f"[bench] sequential-exchange leg skipped: {exc}", file=sys.stderr)
            finally:
                lat.overlap = True
        rho_sum, phi_sum = lat.mass()
        halo_bytes = getattr(lat, "halo_bytes_per_face", None)
        if use_dist:
            dist.barrier()
        lat.close()
        sites = float(nx) * ny * nz
        per_gpu_sites = sites / world
        kern_ms = dev_ms / a.steps
        achieved = per_gpu_sites * BYTES_PER_LUP / (kern_ms * 1e-3) / 1e9
        schedule = eng_schedule
        workload = f"{nx}x{ny}x{nz} periodic, {a.init} init, " + ("kBT=1e-5 alpha0=0" if a.noise else "zero noise")
        return {
            "value": round(sites * a.steps / wall / 1e6, 1), "ms_per_step": round(wall / a.steps * 1e3, 4),
            "workload": workload, "schedule": schedule, "slab_per_gpu": f"{nx}x{ny}x{nz // world}", "spread": spread,
            "mass_check": [rho_sum, phi_sum], "halo_bytes_per_face": halo_bytes, "halo_transport": transport,
            "halo_overlap": None if seq_ms is None else {"ms_per_step_overlapped": round(wall / a.steps * 1e3, 4),
                                                          "ms_per_step_exchange_after_sweep": round(seq_ms, 4)},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4),
                         "traffic": load_traffic(f"{sx}x{sy}x{sz}" + (" noise" if a.noise else ""), schedule),
                         "kernel": "all kernels of one step (hipEvent time / steps)",
                         "algorithmic_bytes_per_launch": per_gpu_sites * BYTES_PER_LUP,
                         "avg_launch_ms": round(kern_ms, 4),
                         "kernel_ms_from_profiles": load_traffic(f"{sx}x{sy}x{sz}" + (" noise" if a.noise else ""), schedule, "kernel_avg_ms")},
        }

    # Workload.  The 60 % target of BASELINE.json is quoted on a 512^3 step at 1 GPU, so that is the headline
    # (81.6 GB of populations); configs[1]'s 256^3 is timed in the same run and reported under config.also.
    # N > 1: weak scaling with the same 512x512x512 slab per GPU.  --size / --shape select one explicit case.
    explicit = bool(a.shape) or a.size > 0
    if a.shape:
        head = tuple(int(v) for v in a.shape.split(","))
    else:
        S = a.size if a.size > 0 else 512
        head = (S, S, S)
    also = {}
    if not explicit and world == 1:
        r = run_case(256, 256, 256)
        also[r["workload"]] = {k: r[k] for k in ("value", "ms_per_step", "spread", "schedule", "roofline")}
    res = run_case(*head)

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
                       "halo_overlap": res["halo_overlap"], "also": also or None},
            "roofline": res["roofline"],
        }
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(noise=a.noise)
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)
    if use_dist:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
