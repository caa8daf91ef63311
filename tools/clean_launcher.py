"""A process that never touches the GPU and starts other programs on request.

On the GPU pool a process that has initialised the GPU must not exec another program (not even in a forked child), so a test
session that has already run GPU tests cannot launch ``torch.distributed.run`` itself.  tests/conftest.py starts this helper at
session start -- before anything has initialised the GPU -- and sends it one JSON line per launch on stdin:
    {"cmd": [...], "env": {...}, "cwd": "...", "timeout": seconds}
The reply is one JSON line on stdout: {"returncode": int, "stdout": str, "stderr": str}.  EOF on stdin ends the helper."""
import json
import subprocess
import sys

for line in sys.stdin:
    line = line.strip()
    if not line:
        continue
    try:
        req = json.loads(line)
        out = subprocess.run(req["cmd"], env=req.get("env"), cwd=req.get("cwd"), capture_output=True, text=True,
                             timeout=req.get("timeout", 600))
        rep = {"returncode": out.returncode, "stdout": out.stdout[-20000:], "stderr": out.stderr[-20000:]}
    except subprocess.TimeoutExpired as e:
        rep = {"returncode": -9, "stdout": (e.stdout or b"").decode("utf-8", "replace")[-20000:] if isinstance(e.stdout, bytes) else (e.stdout or ""),
               "stderr": "timeout"}
    except Exception as e:  # report, keep serving
        rep = {"returncode": -1, "stdout": "", "stderr": f"{type(e).__name__}: {e}"}
    sys.stdout.write(json.dumps(rep) + "\n")
    sys.stdout.flush()
