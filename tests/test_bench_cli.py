"""bench.py's command line, as far as it can be checked without a GPU."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_more_gpus_than_ranks_is_refused():
    """--gpus N is a statement about the launcher (one rank per GPU through torch.distributed.run), not a request to
    start ranks: alone it is an error, said before anything touches a device."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 2 and "torch.distributed.run with 2 ranks" in r.stderr and r.stdout.strip() == ""
