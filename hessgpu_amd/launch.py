"""Start the ranks of a one-node job as child processes and supervise them.

The reference starts its own workers: one pthread per device (TestWin/MultiThreadSIFT.cpp:231-244) or one
posix_spawn'ed server process per GPU (ServerSiftGPU/ServerSiftGPU.cpp:156-194).  Here a job is one process per
GPU, so `python3 bench.py --gpus N` with no launcher around it comes here: the parent -- which never touches the
GPU, it only spawns (no exec of a process that has initialised HIP) -- starts N children with the rendezvous
environment torch.distributed expects (RANK, LOCAL_RANK, WORLD_SIZE, LOCAL_WORLD_SIZE, MASTER_ADDR = 127.0.0.1,
MASTER_PORT = a free port), relays rank 0's stdout (the one JSON line) to its own, lets the other ranks' stdout
go to stderr, and

  * returns the worst exit code of the ranks (a rank ended by signal s counts as 128 + s);
  * once any rank has ended non-zero, gives the others `grace_s` to follow (they may be about to fail the same
    way and say why), then ends their process groups: SIGTERM, and SIGKILL `kill_after_s` later -- a rank blocked
    in a collective on a dead peer would otherwise sit there for the backend's 30-minute timeout;
  * ends every rank the same way when `timeout_s` passes (exit code 124, as timeout(1));
  * ends every rank when it is itself told to stop (SIGTERM / SIGINT), and has the kernel SIGKILL them if it
    dies without the chance (PR_SET_PDEATHSIG).

Every rank is the leader of its own session, so a rank's helper processes go with it.
"""
import ctypes
import os
import signal
import socket
import subprocess
import sys
import threading
import time

EXIT_TIMEOUT = 124


def free_port(addr="127.0.0.1"):
    s = socket.socket()
    try:
        s.bind((addr, 0))
        return s.getsockname()[1]
    finally:
        s.close()


try:   # (loaded in the parent: the child between fork and exec then only makes the one system call)
    _LIBC = ctypes.CDLL(None, use_errno=True)
except Exception:
    _LIBC = None


def _child_setup():
    # runs in the child between fork and exec (Popen has done setsid): have the kernel end this rank if the launcher dies
    if _LIBC is not None:
        _LIBC.prctl(1, int(signal.SIGKILL), 0, 0, 0)   # PR_SET_PDEATHSIG


def rank_env(rank, world, port, base=None, addr="127.0.0.1"):
    """Environment of rank `rank` of a one-node job of `world` ranks."""
    env = dict(os.environ if base is None else base)
    env.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world),
               GROUP_RANK="0", MASTER_ADDR=addr, MASTER_PORT=str(port), HESS_LAUNCHED_BY="hessgpu_amd.launch")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: the only kind the host driver supports
    env.setdefault("OMP_NUM_THREADS", "1")              # as torch.distributed.run does for N > 1
    return env


def _end_group(p, sig):
    try:
        os.killpg(p.pid, sig)
    except (ProcessLookupError, PermissionError):
        pass


def _code(rc):
    return 128 - rc if rc < 0 else rc


def run_ranks(cmd, world, timeout_s=None, grace_s=5.0, kill_after_s=5.0, env=None, out=None, err=None, poll_s=0.05):
    """Run `cmd` (argv list) as ranks 0 .. world-1; -> worst exit code (0 = every rank ended 0)."""
    out = sys.stdout if out is None else out
    err = sys.stderr if err is None else err
    port = free_port()
    procs, relay = [], None
    stop = {"sig": None}

    def on_signal(signum, _frame):
        stop["sig"] = signum

    old = {}
    if threading.current_thread() is threading.main_thread():
        for s in (signal.SIGTERM, signal.SIGINT):
            old[s] = signal.signal(s, on_signal)
    try:
        err_fd = err.fileno() if hasattr(err, "fileno") else 2
        for r in range(world):
            procs.append(subprocess.Popen(cmd, env=rank_env(r, world, port, env), stdin=subprocess.DEVNULL,
                                          stdout=subprocess.PIPE if r == 0 else err_fd, stderr=err_fd,
                                          start_new_session=True, preexec_fn=_child_setup))

        def pump(src):
            for line in iter(src.readline, b""):
                out.write(line.decode(errors="replace"))
                out.flush()

        relay = threading.Thread(target=pump, args=(procs[0].stdout,), daemon=True)
        relay.start()
        t0 = time.monotonic()
        first_bad = None       # time the first non-zero exit was seen
        term_at = None         # time SIGTERM went out
        killed = False
        verdict = None
        while True:
            rcs = [p.poll() for p in procs]
            if all(rc is not None for rc in rcs):
                break
            now = time.monotonic()
            end_now = False
            if term_at is not None:
                if not killed and now - term_at >= kill_after_s:
                    killed = True
                    for p in procs:
                        if p.poll() is None:
                            _end_group(p, signal.SIGKILL)
            elif stop["sig"] is not None:
                verdict, end_now = 128 + stop["sig"], True
                print(f"launch: signal {stop['sig']}: ending the ranks", file=err)
            elif timeout_s is not None and now - t0 >= timeout_s:
                verdict, end_now = EXIT_TIMEOUT, True
                print(f"launch: {timeout_s:.0f} s passed: ending the ranks", file=err)
            elif any(rc not in (None, 0) for rc in rcs):
                if first_bad is None:
                    first_bad = now
                    bad = [(r, _code(rc)) for r, rc in enumerate(rcs) if rc not in (None, 0)]
                    print(f"launch: rank(s) ended non-zero {bad}: the others have {grace_s:.0f} s to follow", file=err)
                elif now - first_bad >= grace_s:
                    end_now = True
            if end_now:
                term_at = now
                for p in procs:
                    if p.poll() is None:
                        _end_group(p, signal.SIGTERM)
            time.sleep(poll_s)
        relay.join(timeout=5.0)
        codes = [_code(p.returncode) for p in procs]
        if verdict is not None:
            return verdict
        if first_bad is not None:
            # the ranks that were ended BY the launcher (SIGTERM / SIGKILL) are a consequence: report the cause
            own = [c for c in codes if c not in (0, 128 + signal.SIGTERM, 128 + signal.SIGKILL)]
            return max(own) if own else max(codes)
        return max(codes)
    finally:
        for p in procs:
            if p.poll() is None:
                _end_group(p, signal.SIGKILL)
        for p in procs:
            try:
                p.wait(timeout=5.0)
            except Exception:
                pass
            if p.stdout is not None:
                try:
                    p.stdout.close()
                except Exception:
                    pass
        for s, h in old.items():
            signal.signal(s, h)


def main(argv=None):
    """python -m hessgpu_amd.launch N [--timeout S] -- program args..."""
    argv = list(sys.argv[1:] if argv is None else argv)
    if "--" not in argv or not argv or not argv[0].isdigit():
        print(main.__doc__, file=sys.stderr)
        return 2
    cut = argv.index("--")
    head, cmd = argv[:cut], argv[cut + 1:]
    timeout = float(head[head.index("--timeout") + 1]) if "--timeout" in head else None
    return run_ranks(cmd, int(head[0]), timeout_s=timeout)


if __name__ == "__main__":
    sys.exit(main())
