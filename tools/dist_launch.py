#!/usr/bin/env python
"""One-node launcher: starts N fresh rank processes of a script, one per GPU, torchrun-style env (RANK / LOCAL_RANK /
WORLD_SIZE / MASTER_ADDR / MASTER_PORT).  Replaces `python -m torch.distributed.launch` in the reference's
tools/dist_train.sh:9-17 for the case where nothing else wraps the script; `bench.py --gpus N` uses it when it is started
as a plain `python bench.py` (no WORLD_SIZE in the env).

This module imports NOTHING that could touch the GPU (no torch): the parent only forks children, relays rank 0's stdout and
returns the worst exit code.  The children are new interpreters (subprocess, not os.exec* of a process that has initialised
HIP), so the rule "never replace a running program from a process that has initialised the GPU" holds by construction.

    python tools/dist_launch.py --nproc 8 [--port P] SCRIPT [ARGS ...]
"""
import os
import signal
import socket
import subprocess
import sys
import threading


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _parse_cpulist(txt):
    out = set()
    for part in txt.strip().split(","):
        if not part:
            continue
        a, _, b = part.partition("-")
        out.update(range(int(a), int(b or a) + 1))
    return out


def gpu_local_cpus():
    """Per GPU (PCI bus order, which is the order HIP numbers devices in when HIP_VISIBLE_DEVICES is unset): the set of host CPUs on
    the GPU's NUMA node, from /sys/class/drm/card*/device/local_cpulist.  [] when sysfs does not say (no GPU, a container)."""
    import glob
    cards = []
    for d in glob.glob("/sys/class/drm/card[0-9]*/device"):
        try:
            if open(os.path.join(d, "vendor")).read().strip() != "0x1002":   # AMD
                continue
            cpus = _parse_cpulist(open(os.path.join(d, "local_cpulist")).read())
            cards.append((os.path.basename(os.path.realpath(d)), cpus))
        except (OSError, ValueError):
            continue
    cards.sort()
    return [c for _, c in cards]


def rank_cpus(world, allowed=None, gpu_cpus=None):
    """CPU set of every local rank: disjoint, each near its GPU.  A train step enqueues several hundred launches from ONE Python thread
    per rank; two ranks time-sharing a core (or a rank migrating across NUMA nodes) makes that rank host-bound, and data-parallel
    training runs at the pace of the slowest rank.  Ranks whose GPUs share a NUMA node split that node's allowed cores evenly; without
    topology information (or when a node has fewer cores than ranks) the allowed cores are split evenly in order."""
    allowed = sorted(os.sched_getaffinity(0) if allowed is None else allowed)
    gpu_cpus = gpu_local_cpus() if gpu_cpus is None else gpu_cpus
    out = [None] * world
    if len(gpu_cpus) >= world:
        groups = {}
        for r in range(world):
            groups.setdefault(frozenset(gpu_cpus[r]), []).append(r)
        ok = True
        for node, ranks in groups.items():
            cores = [c for c in allowed if c in node]
            per = len(cores) // len(ranks)
            if per < 1:
                ok = False
                break
            for i, r in enumerate(ranks):
                out[r] = cores[i * per:(i + 1) * per]
        if ok:
            return out
    per = len(allowed) // world
    if per < 1:       # more ranks than cores: no pinning (sharing cannot be avoided, let the scheduler balance)
        return [list(allowed) for _ in range(world)]
    return [allowed[r * per:(r + 1) * per] for r in range(world)]


def rank_env(rank, world, port, base=None, addr="127.0.0.1", cpus=None):
    env = dict(os.environ if base is None else base)
    env.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world),
               MASTER_ADDR=addr, MASTER_PORT=str(port))
    ncpu = len(cpus) if cpus else max(1, (os.cpu_count() or 8) // max(world, 1))
    # one Python thread drives each GPU; a wide OpenMP pool per rank only fights the other ranks for the host cores
    env.setdefault("OMP_NUM_THREADS", str(max(1, min(8, ncpu))))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: the only form this image's driver supports (RCCL needs it)
    if cpus:
        env["VFMSEG_RANK_CPUS"] = ",".join(str(c) for c in cpus)   # (for the rank's own diagnostics; the mask itself is set before exec)
    return env


def launch(cmd, world, port=None, relay_stdout=None, timeout=None):
    """Run `cmd` (argv list) as `world` rank processes.  Rank 0's stdout is relayed line by line to `relay_stdout`
    (default: this process's stdout); the other ranks' stdout goes to stderr with a rank prefix; stderr is inherited.
    Returns the first non-zero exit code (0 when every rank exited 0).  If one rank dies the others are terminated:
    a collective with a missing peer would otherwise hang until its own timeout."""
    port = port or int(os.environ.get("MASTER_PORT", 0)) or free_port()
    out = relay_stdout or sys.stdout
    procs = []
    pin = os.environ.get("VFMSEG_PIN_RANKS", "1") != "0" and world > 1 and hasattr(os, "sched_setaffinity")
    cpus = rank_cpus(world) if pin else [None] * world
    for r in range(world):
        # the affinity mask is set in the child between fork and exec: before the interpreter, torch and the HIP runtime start
        # their threads, which inherit it
        pre = (lambda c=cpus[r]: os.sched_setaffinity(0, c)) if cpus[r] else None
        procs.append(subprocess.Popen(cmd, env=rank_env(r, world, port, cpus=cpus[r]), stdout=subprocess.PIPE, stderr=None, text=True, bufsize=1,
                                      preexec_fn=pre))

    def pump(r, p):
        for line in p.stdout:
            if r == 0:
                out.write(line)
                out.flush()
            else:
                sys.stderr.write(f"[rank {r}] {line}")
                sys.stderr.flush()

    threads = [threading.Thread(target=pump, args=(r, p), daemon=True) for r, p in enumerate(procs)]
    for t in threads:
        t.start()
    code = 0
    try:
        pending = set(range(world))
        import time
        t0 = time.time()
        while pending:
            for r in list(pending):
                rc = procs[r].poll()
                if rc is None:
                    continue
                pending.discard(r)
                if rc != 0 and code == 0:
                    code = rc
                    sys.stderr.write(f"[dist_launch] rank {r} exited with code {rc}; stopping the other ranks\n")
                    for q in pending:
                        procs[q].terminate()
            if timeout is not None and time.time() - t0 > timeout and pending:
                sys.stderr.write(f"[dist_launch] timeout after {timeout} s; stopping ranks {sorted(pending)}\n")
                code = code or 124
                for q in pending:
                    procs[q].terminate()
                timeout = None
            time.sleep(0.05)
    except KeyboardInterrupt:
        for p in procs:
            if p.poll() is None:
                p.send_signal(signal.SIGINT)
        code = 130
    finally:
        for p in procs:
            try:
                p.wait(timeout=15)
            except subprocess.TimeoutExpired:
                p.kill()
        for t in threads:
            t.join(timeout=5)
    return code


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    world, port = 1, None
    while argv and argv[0].startswith("--"):
        k = argv.pop(0)
        if k in ("--nproc", "--nproc-per-node", "--nproc_per_node"):
            world = int(argv.pop(0))
        elif k.startswith("--nproc") and "=" in k:
            world = int(k.split("=", 1)[1])
        elif k in ("--port", "--master-port", "--master_port"):
            port = int(argv.pop(0))
        elif k.startswith("--master") and "=" in k:
            if "port" in k:
                port = int(k.split("=", 1)[1])
        elif k == "--":
            break
        else:
            raise SystemExit(f"dist_launch: unknown option {k}")
    if not argv:
        raise SystemExit(__doc__)
    return launch([sys.executable] + argv, world, port)


if __name__ == "__main__":
    sys.exit(main())
