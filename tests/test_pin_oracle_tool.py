"""tools/pin_oracle.py must say LOUDLY that nothing was pinned when the reference's environment (TensorFlow 2.0) is
absent -- and never pretend otherwise (exit code 3, "PARITY REMAINS UNPINNED" on stderr)."""
import importlib.util
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_pin_tool_skips_loudly_without_tensorflow():
    if importlib.util.find_spec("tensorflow") is not None:          # then the tool really runs; not this test's business
        return
    res = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pin_oracle.py")], capture_output=True, text=True,
                         timeout=120)
    assert res.returncode == 3
    assert "PARITY REMAINS UNPINNED" in res.stderr and "tensorflow" in res.stderr
    assert "PINNED (" not in res.stdout
