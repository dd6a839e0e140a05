"""torch.distributed.run on a free local port, re-tried on a port collision.

"bind to port 0, read the number, close, hand it to torchrun" leaves a window in which something else on the box takes the
port (seen once in a full GPU suite: `EADDRINUSE` from the rendezvous store): a collision is not a test failure, so the
launch is repeated on another port."""
import socket
import subprocess
import sys
from typing import Dict, List, Optional


def free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def run_torchrun(nproc: int, script_and_args: List[str], env: Optional[Dict[str, str]] = None, cwd: Optional[str] = None,
                 timeout: int = 600, attempts: int = 4) -> subprocess.CompletedProcess:
    res = None
    for _ in range(attempts):
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr",
               "127.0.0.1", "--master-port", str(free_port())] + list(script_and_args)
        res = subprocess.run(cmd, env=env, cwd=cwd, capture_output=True, text=True, timeout=timeout)
        if res.returncode == 0 or not ("EADDRINUSE" in res.stderr or "address already in use" in res.stderr.lower()):
            break
    return res


def init_world1_process_group(backend: str, **kw) -> None:
    """`dist.init_process_group(world_size=1)` on a free local port (env rendezvous), re-tried on a port collision."""
    import os
    import torch.distributed as dist
    last = None
    for _ in range(4):
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()))
        try:
            dist.init_process_group(backend=backend, rank=0, world_size=1, **kw)
            return
        except Exception as e:                    # noqa: BLE001 -- DistNetworkError / RuntimeError depending on the torch build
            if "EADDRINUSE" not in str(e) and "address already in use" not in str(e).lower():
                raise
            last = e
    raise last
