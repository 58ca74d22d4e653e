"""distributed.Watchdog: a multi-rank phase that never finishes (a collective captured on one rank
and not on another hangs inside the runtime, it does not raise) must end the process with a
non-zero status instead of blocking the launcher -- without re-executing anything."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(code):
    return subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True,
                          timeout=120)


def test_watchdog_ends_a_hung_phase_with_status_124():
    r = _run("import time\n"
             "from tensorflowraytrace_amd.distributed import Watchdog\n"
             "with Watchdog(0.5, 'the first multi-rank step', only_distributed=False):\n"
             "    time.sleep(60)\n"
             "print('not reached')\n")
    assert r.returncode == 124
    assert "did not finish within" in r.stderr and "not reached" not in r.stdout


def test_watchdog_is_silent_when_the_phase_finishes_and_off_for_one_process():
    r = _run("import time\n"
             "from tensorflowraytrace_amd.distributed import Watchdog\n"
             "with Watchdog(30, 'quick', only_distributed=False):\n"
             "    pass\n"
             "with Watchdog(0.2, 'single process: no timer'):\n"
             "    time.sleep(0.6)\n"
             "print('done')\n")
    assert r.returncode == 0 and "done" in r.stdout
