"""Command runner for the GPU tests that need a FRESH process (RCCL process groups, bench.py under torchrun).

tests/conftest.py starts this helper at session start, i.e. before the pytest process has initialised the GPU.  The helper
never touches the GPU itself; it only forks + execs the commands it is sent.  That keeps every exec out of processes that
hold a GPU context (on the GPU pool an exec from such a process is refused).  Protocol: one JSON object per line on stdin
({"cmd": [...], "env": {...}, "cwd": ..., "timeout": s}), one per line on stdout ({"rc": int, "out": str})."""
import json
import subprocess
import sys


def main():
    for line in sys.stdin:
        line = line.strip()
        if not line:
            continue
        req = json.loads(line)
        try:
            p = subprocess.run(req["cmd"], env=req.get("env"), cwd=req.get("cwd"), stdout=subprocess.PIPE,
                               stderr=subprocess.STDOUT, text=True, timeout=req.get("timeout", 600))
            rep = {"rc": p.returncode, "out": p.stdout[-20000:]}
        except subprocess.TimeoutExpired as e:
            rep = {"rc": -9, "out": "timeout: %s" % (e.stdout[-4000:] if isinstance(e.stdout, str) else "")}
        except Exception as e:  # noqa: BLE001
            rep = {"rc": -1, "out": repr(e)}
        sys.stdout.write(json.dumps(rep) + "\n")
        sys.stdout.flush()


if __name__ == "__main__":
    main()
